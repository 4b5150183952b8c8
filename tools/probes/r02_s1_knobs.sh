#!/bin/bash
# sweep-1 experiments on the 30-qubit QFT (GPU box): lane-gate paths
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02j
mkdir -p $O
export QH_RELAYOUT=0
for v in "QH_LANE_VALU=1" "QH_LANE_VALU=2" "QH_LANE_VALU=0" "QH_LTAB_LDS=0" "QH_WAVE_BITS=0" "QH_WAVE_BITS=2"; do
  echo "== $v" >> $O/knobs.txt
  env $v bash $R/tools/trace_sweeps.sh >> $O/knobs.txt 2>&1
done
cat $O/knobs.txt
