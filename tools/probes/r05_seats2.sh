#!/bin/bash
# round 5: lane seats by cost, second A/B (the relayout store now orders the lane-resident bits by index bit, not by seat)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05seats2; mkdir -p $O
cd $R
for round in 1 2 3; do for v in off on; do for w in sup30 sup30s1 sup30s2 sup30s3 sup30s5 sup30s7; do
  echo "## $v $w round $round" >> $O/ab.txt
  if [ $v = off ]; then export QH_SEATS=0; else unset QH_SEATS; fi
  QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 5 2>&1 | grep -a "qh sweeps" | tail -4 >> $O/ab.txt
done; done; done
python3 - <<'PY' > gpurun_out/r05seats2/summary.txt
import re, collections, statistics
cur=None; per=collections.defaultdict(list)
for l in open('gpurun_out/r05seats2/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        per[cur].append([float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])])
for k in sorted(per, key=lambda k:(k[1],k[0])):
    pp=per[k]; tot=[sum(p) for p in pp]
    print('%-4s %-9s n %2d total median %7.3f min %7.3f | per sweep median'%(k[0],k[1],len(pp),statistics.median(tot),min(tot)), [round(statistics.median(x),3) for x in zip(*pp)])
PY
cat $O/summary.txt
tools/membench/lowbits 30 8 > $R/gpurun_out/r05_lowbits.txt 2>&1; cat $R/gpurun_out/r05_lowbits.txt
