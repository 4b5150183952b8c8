#!/bin/bash
# round 4, GPU session K: units per wave by workgroup shape (default: 4 for four-wave super-tiles) vs off, on every workload;
# then supremacy-30 with smaller register tiles
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04k; mkdir -p $O
cd $R
for round in 1 2 3; do for u in 1 0 1 0; do for w in qft30 sup30 qft30c64 qft31 qft32 qft33 grover34; do
  echo "## units=$u $w round $round" >> $O/units.txt
  if [ $u = 0 ]; then
    QH_SWEEP_TIMING=1 timeout 400 python tools/run_workload.py $w 3 2>&1 | grep -a "qh sweeps\|step ms" | tail -3 >> $O/units.txt
  else
    QH_SWEEP_UNITS=1 QH_SWEEP_TIMING=1 timeout 400 python tools/run_workload.py $w 3 2>&1 | grep -a "qh sweeps\|step ms" | tail -3 >> $O/units.txt
  fi
done; done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list); per=collections.defaultdict(list)
for l in open('gpurun_out/r04k/units.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        data[cur].append(sum(v)); per[cur].append(v)
for k in sorted(data, key=lambda k:(k[1],k[0])):
    n=len(per[k][0]); pp=[p for p in per[k] if len(p)==n]
    print(k, '(units=0: default heuristic)', 'median total ms %.3f  min %.3f  n %d'%(statistics.median(data[k]),min(data[k]),len(data[k])), 'per sweep median', [round(statistics.median(x),3) for x in zip(*pp)])
PY
bash tools/probes/r04_j.sh > /dev/null 2>&1
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list)
for l in open('gpurun_out/r04j/rb.txt'):
    if l.startswith('##'): cur=l.split(' round')[0][3:]
    elif 'step ms' in l: data[cur]+=[float(x) for x in l.split('step ms')[1].split()][1:]
for k,v in data.items(): print('sup30', k, 'median step ms %.3f min %.3f n %d'%(statistics.median(v), min(v), len(v)))
PY
