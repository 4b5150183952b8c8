#!/bin/bash
# round 4, GPU session O: parity, then interleaved A/B of the library at HEAD against tools/probes/variants/libqcc_hip_base.so
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04o; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_relayout.py tests/test_gpu_sharded.py tests/test_gpu_exchange.py -x -q -m gpu > $O/tests.log 2>&1
grep -a -E "passed|failed|error" $O/tests.log | tail -3
W=${WORKLOADS:-"sup30 qft30 qft33 grover34"}
for round in 1 2 3; do for v in new base; do for w in $W; do
  echo "## $v $w round $round" >> $O/ab.txt
  case $v in
    base) export QCC_HIP_LIB=$R/tools/probes/variants/libqcc_hip_base.so; unset QH_LTAB_ISLAND;;
    new) unset QCC_HIP_LIB; unset QH_LTAB_ISLAND;;
    new_ltab) unset QCC_HIP_LIB; export QH_LTAB_ISLAND=1;;
  esac
  QH_SWEEP_TIMING=1 timeout 400 python tools/run_workload.py $w 4 2>&1 | grep -a "qh sweeps\|step ms" | tail -3 >> $O/ab.txt
done; done; done
unset QCC_HIP_LIB QH_LTAB_ISLAND
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list); per=collections.defaultdict(list)
for l in open('gpurun_out/r04o/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        data[cur].append(sum(v)); per[cur].append(v)
for k in sorted(data, key=lambda k:(k[1],k[0])):
    n=len(per[k][0]); pp=[p for p in per[k] if len(p)==n]
    print(k, 'median total ms %.3f  min %.3f  n %d'%(statistics.median(data[k]),min(data[k]),len(data[k])), 'per sweep median', [round(statistics.median(x),3) for x in zip(*pp)])
PY
