#!/bin/bash
# round 4, GPU session J: supremacy-30 with smaller register tiles (more waves per SIMD: is the op stream latency-bound?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j; mkdir -p $O
cd $R
for round in 1 2 3; do for v in "X=0" "QH_SWEEP_RB=4" "QH_SWEEP_RB=4 QH_WAVE_BITS=2" "QH_SWEEP_RB=4 QH_WAVE_BITS=1" "QH_WAVE_BITS=1" "QH_SWEEP_RB=3 QH_WAVE_BITS=2"; do
  echo "## $v round $round" >> $O/rb.txt
  env $v QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py sup30 4 2>&1 | grep -a "qh sweeps\|step ms" | tail -3 >> $O/rb.txt
done; done
cat $O/rb.txt
