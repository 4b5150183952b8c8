#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "QH_SWEEP_GRID=0" "QH_SWEEP_GRID=1536" "QH_SWEEP_GRID=3072" "QH_SWEEP_GRID=6144" "QH_SWEEP_GRID=24576"; do
  echo "== $v"
  env $v bash $R/tools/trace_sweeps.sh 2>&1 | tail -1
done
