#!/bin/bash
# round 5: lane SEATS by cost (planner.h choose_seats): parity first, then an interleaved A/B on the op-heavy workloads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05seats; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_relayout.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -5 > $O/pytest.txt
QH_SEATS=2 timeout 400 python tools/fuzz_parity.py 240 501 2>&1 | tail -4 > $O/fuzz_seats2.txt
timeout 300 python tools/fuzz_parity.py 150 502 2>&1 | tail -4 > $O/fuzz_default.txt
for round in 1 2 3 4; do for v in off on; do for w in sup30 sup30s1 sup30s7 qft30 qft33; do
  echo "## $v $w round $round" >> $O/ab.txt
  if [ $v = off ]; then export QH_SEATS=0; else unset QH_SEATS; fi
  QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 5 2>&1 | grep -a "qh sweeps" | tail -4 >> $O/ab.txt
done; done; done
python3 - <<'PY' > gpurun_out/r05seats/summary.txt
import re, collections, statistics
cur=None; per=collections.defaultdict(list)
for l in open('gpurun_out/r05seats/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        per[cur].append([float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])])
for k in sorted(per, key=lambda k:(k[1],k[0])):
    pp=per[k]; tot=[sum(p) for p in pp]
    print('%-4s %-9s n %2d total median %7.3f min %7.3f | per sweep median'%(k[0],k[1],len(pp),statistics.median(tot),min(tot)), [round(statistics.median(x),3) for x in zip(*pp)])
PY
cat $O/pytest.txt $O/fuzz_seats2.txt $O/fuzz_default.txt $O/summary.txt
