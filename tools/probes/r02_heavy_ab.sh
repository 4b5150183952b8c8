#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
for v in "QH_HEAVY_SWEEP_OPS=11" "QH_HEAVY_SWEEP_OPS=99"; do
  echo "== $v"
  env $v bash $R/tools/trace_sweeps.sh 2>&1 | tail -1
  env $v python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-ladder-base --no-cached-plan 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
done
