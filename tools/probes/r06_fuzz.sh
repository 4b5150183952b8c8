#!/bin/bash
# round 6: GPU fuzz of the round's final build (plan_best: the level search as a portfolio of host threads, predicted-time choice): fused path vs oracle
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06fuzz; mkdir -p $O
cd $R
B=${1:-16000}     # seed base: another base, another set of circuits
T=${2:-300}
(timeout $((T + 100)) python tools/fuzz_parity.py $T $((B + 1)) > $O/f_default.txt 2>&1) &
(QH_PLAN_SEARCH_STEPS=300000 timeout $((T + 100)) python tools/fuzz_parity.py $T $((B + 2)) > $O/f_search.txt 2>&1) &
(QH_PLAN_SEARCH_STEPS=300000 QH_PLAN_ONLY_WB=2 QH_SEATS=2 timeout $((T + 100)) python tools/fuzz_parity.py $T $((B + 3)) > $O/f_search_onlywb2_seats2.txt 2>&1) &
(QH_PLAN_SEARCH_STEPS=300000 QH_RELAYOUT=0 timeout $((T + 100)) python tools/fuzz_parity.py $T $((B + 4)) > $O/f_search_inplace.txt 2>&1) &
(QH_PLAN_SEARCH_STEPS=300000 FUZZ_BW=64 timeout $((T + 100)) python tools/fuzz_parity.py $T $((B + 5)) > $O/f_search_c64.txt 2>&1) &
(QH_SEATS=2 QH_WAVE_BITS=2 QH_LANE_VALU=2 timeout $((T + 100)) python tools/fuzz_parity.py $T $((B + 6)) > $O/f_seats2_wb2_valu2.txt 2>&1) &
wait
tail -n 2 $O/f_*.txt
