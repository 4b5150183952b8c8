#!/bin/bash
# round 4, GPU session E: group classes in the linear island layout vs the build before; per-op timelines of both
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_lib.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
tail -4 $O/pytest_subset.log
for round in 1 2 3; do for lib in before classes3; do for w in sup30 qft30 qft33 qft30c64; do
  echo "## $lib $w round $round" >> $O/ab.txt
  QCC_HIP_LIB=$R/tools/probes/variants/libqcc_$lib.so QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 4 2>&1 | grep -a "qh sweeps\|step ms" | tail -4 >> $O/ab.txt
done; done; done
for lib in before classes3; do for w in qft30 sup30; do
  rm -f /tmp/prof.txt
  QH_PLAN_CACHE=0 QH_PROF_OUT=/tmp/prof.txt QCC_HIP_LIB=$R/tools/probes/variants/libqcc_prof_$lib.so timeout 300 python3 tools/run_workload.py $w 1 > /dev/null 2>&1
  python3 - /tmp/prof.txt > $O/op_timeline_${w}_$lib.txt <<'PY'
import sys
txt = open(sys.argv[1]).read().split('sweep 0 ')
print('sweep 0 ' + txt[-1])
PY
done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list); per=collections.defaultdict(list)
for l in open('gpurun_out/r04e/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        data[cur].append(sum(v)); per[cur].append(v)
for k in sorted(data, key=lambda k:(k[1],k[0])):
    n=len(per[k][0]); pp=[p for p in per[k] if len(p)==n]
    print(k, 'median total ms %.3f  min %.3f  n %d'%(statistics.median(data[k]),min(data[k]),len(data[k])), 'per sweep', [round(statistics.median(x),3) for x in zip(*pp)])
PY
