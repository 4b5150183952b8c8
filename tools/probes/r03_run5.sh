cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_sharded.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r03e/pytest.log
cat gpurun_out/r03e/pytest.log
# GPU fuzz: random circuits through the fused path vs the oracle, every plan shape (tools/fuzz_parity.py)
for seed in 301 302; do timeout 400 python tools/fuzz_parity.py 200 $seed 2>&1 | tail -3; done > gpurun_out/r03e/fuzz.log 2>&1
QH_PLAN_SEARCH_STEPS=300000 timeout 400 python tools/fuzz_parity.py 200 303 2>&1 | tail -3 >> gpurun_out/r03e/fuzz.log
QH_WAVE_BITS=2 QH_LANE_VALU=2 timeout 400 python tools/fuzz_parity.py 150 304 2>&1 | tail -3 >> gpurun_out/r03e/fuzz.log
FUZZ_BW=64 timeout 400 python tools/fuzz_parity.py 150 305 2>&1 | tail -3 >> gpurun_out/r03e/fuzz.log
cat gpurun_out/r03e/fuzz.log
