#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "QH_SWEEP_ROT=3" "QH_SWEEP_ROT=0" "QH_SWEEP_ROT=2" "QH_SWEEP_ROT=4" "QH_SUPERS_PER_BLOCK=1" "QH_SWEEP_BLOCK_WAVES=2" "QH_LTAB_LDS=8" "QH_RELAYOUT_AHEAD=0"; do
  echo "== $v"
  env $v bash $R/tools/trace_sweeps.sh 2>&1 | tail -1
done
