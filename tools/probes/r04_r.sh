#!/bin/bash
# round 4, GPU session R: op timelines of sampled waves with the argument block (kernel entry -> island, tile load, store issue);
# the QFT also with the lane tables fetched by the island
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04r; mkdir -p $O
cd $R
bash tools/probes/prof_island.sh run qft30 $O/op_timeline_qft30.txt > /dev/null 2>&1
QH_LTAB_ISLAND=1 bash tools/probes/prof_island.sh run qft30 $O/op_timeline_qft30_ltab_island.txt > /dev/null 2>&1
bash tools/probes/prof_island.sh run sup30 $O/op_timeline_sup30.txt > /dev/null 2>&1
for f in $O/op_timeline_qft30.txt $O/op_timeline_qft30_ltab_island.txt; do grep "^sweep\|memtime\|tile load\|prologue\|store" $f | tail -18 | cut -c1-200; done
