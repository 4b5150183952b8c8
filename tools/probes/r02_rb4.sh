#!/bin/bash
# Smaller register tiles = more waves per SIMD?  (RB=4: 104 VGPRs, four waves; two wave bits keep a 12-bit tile)
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in sup30 qft30; do
for v in "QH_NOP=1" "QH_SWEEP_RB=4 QH_WAVE_BITS=2" "QH_SWEEP_RB=4 QH_WAVE_BITS=1" "QH_SWEEP_RB=5 QH_WAVE_BITS=2" "QH_SWEEP_RB=5 QH_WAVE_BITS=1" "QH_SWEEP_RB=3 QH_WAVE_BITS=2"; do
  echo "== $w $v"
  env $v bash $R/tools/trace_workload.sh $w 2>&1 | tail -1
done
done
