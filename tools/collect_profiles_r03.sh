#!/bin/bash
# Round-3 evidence (GPU box) -> gpurun_out/r03p/ (copied to profiles/r03/): bench lines (default, driver style, in
# place, unfused, sharded N=1 at 33 qubits through torchrun), kernel statistics over post-warm-up dispatches,
# FETCH/WRITE PMC passes (traffic) for the QFT, supremacy-30 and the complex64 QFT, SQ counters of the new
# 4-sweep supremacy plan, power / clock traces.  ONE session: every figure of DESIGN 7 comes from this run.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ladder-base --no-cached-plan --no-configs"
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-configs > $O/bench_driver_style.json 2> $O/bench_driver_style.err
QH_RELAYOUT=0 timeout 300 python $R/bench.py --no-cpu-baseline --no-ladder-base --no-cached-plan --no-configs > $O/bench_inplace.json 2> $O/bench_inplace.err
timeout 300 python $R/bench.py --fusion 0 --steps 2 --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 $R/bench.py --gpus 1 --sharded --qubits 33 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_sharded33_n1.json 2> $O/bench_sharded33_n1.err
trace() {  # trace <tag> <skip sweeps> <command...>: kernel trace -> steady-state stats
  local tag=$1 skip=$2; shift 2
  rm -rf /tmp/st_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -o s -- "$@" > $O/${tag}_trace.log 2>&1
  f=$(find /tmp/st_$tag -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python3 $R/tools/kernel_stats_steady.py $f $skip $O/${tag}_kernel_stats.csv > /dev/null
  f=$(find /tmp/st_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats_all_dispatches.csv
}
pmc() {    # pmc <tag> <counters> <command...> -> $O/<tag>.csv (counter_collection)
  local tag=$1 ctr=$2; shift 2
  rm -rf /tmp/pm_$tag
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pm_$tag -o p -- "$@" > $O/${tag}.log 2>&1
  f=$(find /tmp/pm_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $O/${tag}.csv
}
per_sweep() {
  python3 - "$1" <<'PY'
import csv, sys, collections
by = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_sweep' in r['Kernel_Name']:
        by.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = int(float(r['Counter_Value']))
for d, c in by.items():
    print(d, c)
PY
}
trace fused 6 $B                                    # 2 warm-up steps x 3 sweeps dropped
trace sup30 8 python $R/tools/run_workload.py sup30 5      # first two circuits dropped
trace qft30c64 6 python $R/tools/run_workload.py qft30c64 5
trace qft33 6 python $R/tools/run_workload.py qft33 4
pmc pmc_fetch_fused FETCH_SIZE $B
pmc pmc_write_fused WRITE_SIZE $B
python3 $R/tools/collect_traffic.py $O/pmc_fetch_fused.csv $O/pmc_write_fused.csv $O/traffic_fused.json > /dev/null
pmc pmc_fetch_unfused FETCH_SIZE python $R/bench.py --fusion 0 --steps 1 --warmup 0 --no-cpu-baseline
pmc pmc_write_unfused WRITE_SIZE python $R/bench.py --fusion 0 --steps 1 --warmup 0 --no-cpu-baseline
python3 $R/tools/collect_traffic.py $O/pmc_fetch_unfused.csv $O/pmc_write_unfused.csv $O/traffic_unfused.json > /dev/null
for w in sup30 qft30c64; do
  pmc pmc_fetch_$w FETCH_SIZE python $R/tools/run_workload.py $w 1
  pmc pmc_write_$w WRITE_SIZE python $R/tools/run_workload.py $w 1
  python3 $R/tools/collect_traffic.py $O/pmc_fetch_$w.csv $O/pmc_write_$w.csv $O/traffic_$w.json > /dev/null
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM"; do
    pmc sq_tmp "$set" python $R/tools/run_workload.py $w 1
    echo "## $set" >> $O/sq_counters_$w.txt
    per_sweep $O/sq_tmp.csv >> $O/sq_counters_$w.txt
  done
  rm -f $O/sq_tmp.csv $O/sq_tmp.log $O/pmc_fetch_$w.csv $O/pmc_write_$w.csv
done
for w in qft30 sup30 qft33; do
  timeout 120 python $R/tools/probes/smi_trace.py $w 20 $O/smi_trace_$w.csv > /dev/null 2>&1
done
rm -f $O/*_trace.log $O/pmc_*.log $O/pmc_fetch_unfused.csv $O/pmc_write_unfused.csv
ls -la $O
for f in bench_default bench_driver_style bench_sharded33_n1; do head -c 1500 $O/$f.json; echo; done
cat $O/fused_kernel_stats.csv $O/sup30_kernel_stats.csv $O/qft30c64_kernel_stats.csv $O/qft33_kernel_stats.csv
cat $O/traffic_fused.json | head -30
