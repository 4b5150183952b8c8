#!/bin/bash
# round 4, GPU session I: the whole GPU suite on the final tree; units per wave (QH_SWEEP_UNITS) A/B in shuffled process order
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
for round in 1 2 3 4; do for u in 1 2 4 2 1 4; do for w in qft30 sup30 qft33; do
  echo "## units=$u $w round $round" >> $O/units.txt
  QH_SWEEP_UNITS=$u QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 4 2>&1 | grep -a "qh sweeps\|step ms" | tail -4 >> $O/units.txt
done; done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list); per=collections.defaultdict(list)
for l in open('gpurun_out/r04i/units.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        data[cur].append(sum(v)); per[cur].append(v)
for k in sorted(data, key=lambda k:(k[1],k[0])):
    n=len(per[k][0]); pp=[p for p in per[k] if len(p)==n]
    print(k, 'median total ms %.3f  min %.3f  n %d'%(statistics.median(data[k]),min(data[k]),len(data[k])), 'per sweep median', [round(statistics.median(x),3) for x in zip(*pp)], 'min', [round(min(x),3) for x in zip(*pp)])
PY
