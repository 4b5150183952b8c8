#!/bin/bash
# round 4, GPU session D: group classes, second cut (no taken branch on the common paths) vs the build before; new tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_lib.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not shard_size" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
tail -4 $O/pytest_subset.log
for round in 1 2 3; do for lib in before classes2; do for w in sup30 qft30 qft33 qft30c64; do
  echo "## $lib $w round $round" >> $O/ab.txt
  QCC_HIP_LIB=$R/tools/probes/variants/libqcc_$lib.so QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 4 2>&1 | grep -a "qh sweeps\|step ms" | tail -4 >> $O/ab.txt
done; done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list); per=collections.defaultdict(list)
for l in open('gpurun_out/r04d/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        data[cur].append(sum(v)); per[cur].append(v)
for k in sorted(data, key=lambda k:(k[1],k[0])):
    n=len(per[k][0]); pp=[p for p in per[k] if len(p)==n]
    print(k, 'median total ms %.3f  min %.3f  n %d'%(statistics.median(data[k]),min(data[k]),len(data[k])), 'per sweep', [round(statistics.median(x),3) for x in zip(*pp)])
PY
