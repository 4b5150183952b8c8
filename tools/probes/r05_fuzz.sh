#!/bin/bash
# round 5: GPU fuzz of the round's final build (seats, argmax from the last sweep, look-ahead on searched tiles): fused path vs oracle
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05fuzz; mkdir -p $O
cd $R
B=${1:-9000}     # seed base: another base, another set of circuits
(timeout 700 python tools/fuzz_parity.py 600 $((B + 1)) > $O/f_default.txt 2>&1) &
(QH_SEATS=2 timeout 700 python tools/fuzz_parity.py 600 $((B + 2)) > $O/f_seats2.txt 2>&1) &
(QH_SEATS=2 QH_WAVE_BITS=2 QH_LANE_VALU=2 timeout 700 python tools/fuzz_parity.py 600 $((B + 3)) > $O/f_seats2_wb2_valu2.txt 2>&1) &
(QH_RELAYOUT=0 timeout 700 python tools/fuzz_parity.py 600 $((B + 4)) > $O/f_inplace.txt 2>&1) &
(QH_PLAN_SEARCH_STEPS=300000 QH_SEATS=2 timeout 700 python tools/fuzz_parity.py 600 $((B + 5)) > $O/f_search_seats2.txt 2>&1) &
(FUZZ_BW=64 timeout 700 python tools/fuzz_parity.py 600 $((B + 6)) > $O/f_c64.txt 2>&1) &
wait
tail -n 3 $O/f_*.txt
