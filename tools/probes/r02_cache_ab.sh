#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
for c in 0 1; do
  QH_PLAN_CACHE=$c python $R/bench.py --steps 10 --warmup 9 --no-cpu-baseline --no-ladder-base --no-cached-plan 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cache $c ms/step', round(d['ms_per_step'],3), 'event', round(d['event_ms_per_step'],3))"
done
done
QH_PLAN_CACHE=1 bash $R/tools/trace_sweeps.sh
QH_PLAN_CACHE=0 bash $R/tools/trace_sweeps.sh
