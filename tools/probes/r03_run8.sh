cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.log; cat $O/pytest.log
for seed in 401 402; do timeout 300 python tools/fuzz_parity.py 150 $seed 2>&1 | tail -2; done > $O/fuzz.log 2>&1
QH_WAVE_BITS=2 QH_LANE_VALU=2 timeout 300 python tools/fuzz_parity.py 120 403 2>&1 | tail -2 >> $O/fuzz.log
FUZZ_BW=64 timeout 300 python tools/fuzz_parity.py 120 404 2>&1 | tail -2 >> $O/fuzz.log
cat $O/fuzz.log
