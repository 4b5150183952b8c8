#!/bin/bash
# round 4, GPU session N: per-op timeline of sampled waves (measurement build of the complex128 RB=5 island) for supremacy-30
# and the 30-qubit QFT at HEAD; the QFT also with the lane tables fetched by the island and with two units per wave
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04n; mkdir -p $O
cd $R
bash tools/probes/prof_island.sh run sup30 $O/op_timeline_sup30.txt > /dev/null 2>&1
bash tools/probes/prof_island.sh run qft30 $O/op_timeline_qft30.txt > /dev/null 2>&1
QH_LTAB_ISLAND=1 bash tools/probes/prof_island.sh run qft30 $O/op_timeline_qft30_ltab_island.txt > /dev/null 2>&1
QH_SWEEP_UNITS=2 bash tools/probes/prof_island.sh run qft30 $O/op_timeline_qft30_units2.txt > /dev/null 2>&1
for f in $O/op_timeline_qft30*.txt; do echo "== $f"; grep "^sweep\|memtime\|tile load\|prologue\|store" $f | tail -18; done
