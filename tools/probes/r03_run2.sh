cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r03b/pytest.log
timeout 600 python bench.py > gpurun_out/r03b/bench_default.json 2> gpurun_out/r03b/bench_default.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --sharded --qubits 33 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03b/bench_sharded33.json 2> gpurun_out/r03b/bench_sharded33.err
cat gpurun_out/r03b/pytest.log; tail -5 gpurun_out/r03b/bench_default.err; head -c 6000 gpurun_out/r03b/bench_default.json; echo; tail -5 gpurun_out/r03b/bench_sharded33.err; head -c 3000 gpurun_out/r03b/bench_sharded33.json
