#!/bin/bash
# round 5: lane seats on / off once more, now at three waves per SIMD (the earlier A/Bs of the round ran at two: profiles/r05/head_vs_r04_ab.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05seats4; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "argmax or readers" 2>&1 | tail -15 > $O/pytest_argmax.txt
for round in 1 2 3; do for v in off on; do for w in sup30 sup30s1 sup30s2 sup30s5 sup30s7; do
  echo "## $v $w round $round" >> $O/ab.txt
  if [ $v = off ]; then export QH_SEATS=0; else unset QH_SEATS; fi
  QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 5 2>&1 | grep -a "qh sweeps" | tail -4 >> $O/ab.txt
done; done; done
unset QH_SEATS
python3 - <<'PY' > gpurun_out/r05seats4/summary.txt
import re, collections, statistics
cur=None; per=collections.defaultdict(list)
for l in open('gpurun_out/r05seats4/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        per[cur].append([float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])])
for k in sorted(per, key=lambda k:(k[1],k[0])):
    pp=per[k]; tot=[sum(p) for p in pp]
    print('%-4s %-9s n %2d total median %7.3f min %7.3f | per sweep median'%(k[0],k[1],len(pp),statistics.median(tot),min(tot)), [round(statistics.median(x),3) for x in zip(*pp)])
PY
cat $O/pytest_argmax.txt $O/summary.txt
python - <<'PY'
import sys, json
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.single_shot(0)))
PY
