cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_relayout.py tests/test_gpu_exchange.py -m gpu -q -x -k "not shard_size" 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
sed -i 's/for w in qft30 qft30c64; do/for w in qft30 qft30c64 sup30 qft33; do/' tools/probes/r03_run_variants.sh
rm -rf gpurun_out/r03v; bash tools/probes/r03_run_variants.sh > $O/variants.log 2>&1; grep -a "median total" $O/variants.log | sort -k3
