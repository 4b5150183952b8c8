#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/$1 (default r01):
# bench lines (fused / unfused), rocprofv3 kernel stats, PMC FETCH/WRITE passes.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py > $O/bench_fused.json 2> $O/bench_fused.err
timeout 300 python $R/bench.py --fusion 0 --steps 2 --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fused -o fused -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_fused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_unfused -o unfused -- python $R/bench.py --fusion 0 --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_unfused.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_u -o fetch -- python $R/bench.py --fusion 0 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch_u.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_u -o write -- python $R/bench.py --fusion 0 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write_u.log 2>&1
timeout 300 python $R/tools/bench_configs.py > $O/bench_configs.json 2>&1
head -c 600 $O/bench_fused.json; echo; ls $O
