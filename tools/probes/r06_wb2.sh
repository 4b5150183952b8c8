#!/bin/bash
# round 6: the repeated tile search with two wave bits (plan_best) -- supremacy-30 seeds 0, 1, 2, 4: sweeps and ms per circuit with
# the new default, with the second-wave-bit search off (round 5's plans), and with two wave bits pinned; interleaved fresh processes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06wb2; mkdir -p $O
cd $R
for round in 1 2; do
  for seed in 0 1 2 4; do
    for v in default wb2off onlywb2; do
      unset QH_PLAN_SEARCH_WB2 QH_PLAN_ONLY_WB
      [ $v = wb2off ] && export QH_PLAN_SEARCH_WB2=0
      [ $v = onlywb2 ] && export QH_PLAN_ONLY_WB=2
      echo "## $v seed $seed round $round" >> $O/ab.txt
      QH_SWEEP_TIMING=1 timeout 120 python tools/run_workload.py sup30s$seed 8 2>&1 | grep -E "step ms|qh sweeps" | tail -4 >> $O/ab.txt
    done
  done
done
unset QH_PLAN_SEARCH_WB2 QH_PLAN_ONLY_WB
python3 - <<'PY'
import re, collections, statistics
rows = collections.defaultdict(list)
key = None
for ln in open('gpurun_out/r06wb2/ab.txt'):
    if ln.startswith('## '):
        p = ln.split(); key = (p[3], p[1])
    elif 'step ms' in ln:
        v = [float(x) for x in ln.split('step ms')[1].split()]
        rows[key].append(statistics.median(v[2:]))
for k in sorted(rows):
    print('seed', k[0], f'{k[1]:8s}', ' '.join(f'{x:.2f}' for x in rows[k]), 'ms per circuit (median of steps 3..8, one figure per process)')
PY
