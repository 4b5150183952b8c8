#!/bin/bash
# round 4, GPU session S: two super-tiles per workgroup (four waves) again, now that the wave start is short (the QH_SUPERS_CONTIG experiment switch was removed afterwards: no gain)
# in-place first sweep (QH_SUPERS_CONTIG=2) and its gather sweeps (QH_SUPERS_PER_BLOCK=2); fresh processes, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04s; mkdir -p $O
cd $R
for round in 1 2 3 4; do for v in default contig2 split2 both2; do for w in qft30 qft30c64 qft31; do
  echo "## $v $w round $round" >> $O/ab.txt
  unset QH_SUPERS_CONTIG QH_SUPERS_PER_BLOCK
  case $v in
    contig2) export QH_SUPERS_CONTIG=2;;
    split2) export QH_SUPERS_PER_BLOCK=2;;
    both2) export QH_SUPERS_CONTIG=2 QH_SUPERS_PER_BLOCK=2;;
  esac
  QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 5 2>&1 | grep -a "qh sweeps" | tail -4 >> $O/ab.txt
done; done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; per=collections.defaultdict(list)
for l in open('gpurun_out/r04s/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        per[cur].append([float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])])
for k in sorted(per, key=lambda k:(k[1],k[0])):
    pp=per[k]; tot=[sum(p) for p in pp]
    print('%-8s %-9s n %2d total median %7.3f min %7.3f | per sweep median'%(k[0],k[1],len(pp),statistics.median(tot),min(tot)), [round(statistics.median(x),3) for x in zip(*pp)], 'min', [round(min(x),3) for x in zip(*pp)])
PY
