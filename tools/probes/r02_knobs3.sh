#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "QH_SWEEP_ROT=4" "QH_SWEEP_ROT=5" "QH_SWEEP_ROT=6" "QH_SWEEP_ROT=8" "QH_SWEEP_ROT=4 QH_SUPERS_PER_BLOCK=1" "QH_SWEEP_ROT=6 QH_SUPERS_PER_BLOCK=1" "QH_SWEEP_ROT=5 QH_RELAYOUT=0" "QH_SWEEP_ROT=3 QH_RELAYOUT=0"; do
  echo "== $v"
  env $v bash $R/tools/trace_sweeps.sh 2>&1 | tail -1
done
