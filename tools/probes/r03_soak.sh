cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s; mkdir -p $O
for seed in 801 802; do timeout 400 python tools/fuzz_parity.py 300 $seed 2>&1 | tail -2; done > $O/fuzz.log 2>&1
QH_PLAN_SEARCH_STEPS=300000 timeout 400 python tools/fuzz_parity.py 300 804 2>&1 | tail -2 >> $O/fuzz.log
FUZZ_BW=64 timeout 400 python tools/fuzz_parity.py 300 805 2>&1 | tail -2 >> $O/fuzz.log
QH_WAVE_BITS=2 QH_LANE_VALU=2 timeout 400 python tools/fuzz_parity.py 200 806 2>&1 | tail -2 >> $O/fuzz.log
QH_LTAB_ISLAND=1 timeout 400 python tools/fuzz_parity.py 200 807 2>&1 | tail -2 >> $O/fuzz.log
QH_ALLOC_CONTIG=1 timeout 400 python tools/fuzz_parity.py 200 808 2>&1 | tail -2 >> $O/fuzz.log
cat $O/fuzz.log
