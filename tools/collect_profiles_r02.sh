#!/bin/bash
# Round-2 evidence (GPU box) -> gpurun_out/r02p/: bench lines, rocprofv3 kernel stats of the bench
# command, FETCH/WRITE PMC passes (traffic), address-translation counters per sweep with and without
# relayout, and kernel stats + traffic + SQ counters of configs 3, 4 and the complex64 QFT.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ladder-base --no-cached-plan"
timeout 600 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
QH_RELAYOUT=0 timeout 300 python $R/bench.py --no-cpu-baseline --no-ladder-base --no-cached-plan > $O/bench_inplace.json 2> $O/bench_inplace.err
timeout 300 python $R/bench.py --fusion 0 --steps 2 --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err
stats() {  # stats <tag> <command...>: rocprofv3 --kernel-trace --stats -> $O/<tag>_kernel_stats.csv
  local tag=$1; shift
  rm -rf /tmp/st_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -o s -- "$@" > $O/${tag}_stats.log 2>&1
  f=$(find /tmp/st_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
}
pmc() {    # pmc <tag> <counters> <command...> -> $O/<tag>.csv (counter_collection)
  local tag=$1 ctr=$2; shift 2
  rm -rf /tmp/pm_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pm_$tag -o p -- "$@" > $O/${tag}.log 2>&1
  f=$(find /tmp/pm_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && cp $f $O/${tag}.csv
}
per_sweep() {   # per_sweep <csv> : counters per k_sweep dispatch
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
stats fused $B
QH_RELAYOUT=0 stats fused_inplace $B
stats unfused python $R/bench.py --fusion 0 --steps 1 --warmup 1 --no-cpu-baseline
pmc pmc_fetch_fused FETCH_SIZE $B
pmc pmc_write_fused WRITE_SIZE $B
python3 $R/tools/collect_traffic.py $O/pmc_fetch_fused.csv $O/pmc_write_fused.csv $O/traffic_fused.json > /dev/null
pmc pmc_fetch_unfused FETCH_SIZE python $R/bench.py --fusion 0 --steps 1 --warmup 0 --no-cpu-baseline
pmc pmc_write_unfused WRITE_SIZE python $R/bench.py --fusion 0 --steps 1 --warmup 0 --no-cpu-baseline
python3 $R/tools/collect_traffic.py $O/pmc_fetch_unfused.csv $O/pmc_write_unfused.csv $O/traffic_unfused.json > /dev/null
# address translation, per sweep launch (steps: warm-up + 3)
for mode in 1 0; do
  for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
             "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; do
    QH_RELAYOUT=$mode pmc tlb_tmp "$set" $B
    echo "## QH_RELAYOUT=$mode  $set" >> $O/translation_counters_qft30.txt
    per_sweep $O/tlb_tmp.csv >> $O/translation_counters_qft30.txt
  done
done
rm -f $O/tlb_tmp.csv $O/tlb_tmp.log
# the other configurations
for w in sup30 qft30c64 grover34; do
  stats $w python $R/tools/run_workload.py $w 2
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
timeout 900 python $R/tools/bench_configs.py > $O/bench_configs.json 2>&1
rm -f $O/*_stats.log $O/pmc_*.log
ls -la $O; head -c 2500 $O/bench_default.json; echo; cat $O/translation_counters_qft30.txt | head -60
# per-op timeline of a wave inside the sweep kernel (measurement build of the island, tools/probes/prof_island.sh)
if [ -f $R/tools/probes/libqcc_hip_prof.so ]; then
  for w in qft30 sup30; do
    bash $R/tools/probes/prof_island.sh run $w /tmp/prof_$w.txt > $O/op_timeline_$w.txt 2>&1
  done
fi
