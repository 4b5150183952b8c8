#!/bin/bash
# round 5: the relayout store's look-ahead uses the tile the search chose for the next sweep (QH_AHEAD_FORCED) and puts the
# bits that tile wants lowest among the lane positions (QH_STORE_ORDER); interleaved A/B, then Grover-34 per sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05seats3; mkdir -p $O
cd $R
for round in 1 2 3; do for v in base ahead ahead_order all_off; do for w in sup30 sup30s1 sup30s2 sup30s5 sup30s7; do
  echo "## $v $w round $round" >> $O/ab.txt
  unset QH_SEATS QH_AHEAD_FORCED QH_STORE_ORDER
  case $v in
    base) export QH_AHEAD_FORCED=0 QH_STORE_ORDER=0;;
    ahead) export QH_STORE_ORDER=0;;
    all_off) export QH_AHEAD_FORCED=0 QH_STORE_ORDER=0 QH_SEATS=0;;
  esac
  QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 5 2>&1 | grep -a "qh sweeps" | tail -4 >> $O/ab.txt
done; done; done
unset QH_SEATS QH_AHEAD_FORCED QH_STORE_ORDER
python3 - <<'PY' > gpurun_out/r05seats3/summary.txt
import re, collections, statistics
cur=None; per=collections.defaultdict(list)
for l in open('gpurun_out/r05seats3/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        per[cur].append([float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])])
for k in sorted(per, key=lambda k:(k[1],k[0])):
    pp=per[k]; tot=[sum(p) for p in pp]
    print('%-11s %-9s n %2d total median %7.3f min %7.3f | per sweep median'%(k[0],k[1],len(pp),statistics.median(tot),min(tot)), [round(statistics.median(x),3) for x in zip(*pp)])
PY
cat $O/summary.txt
QH_SWEEP_TIMING=1 timeout 600 python tools/run_workload.py grover34 4 2>&1 | grep -a "qh sweeps" > $O/grover34_sweeps.txt; cat $O/grover34_sweeps.txt
