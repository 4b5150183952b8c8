#!/bin/bash
# The round's evidence set (GPU box) -> gpurun_out/r${ROUND:-06}p/ (copied to profiles/rNN/): the default bench line (every config, the
# unfused entry, single shot, CPU baseline), the driver-style line, kernel statistics over post-warm-up dispatches for the
# fused QFT, supremacy-30, complex64, 33 qubits, Grover-34 and the PER-GATE kernels, FETCH / WRITE PMC passes (traffic)
# for every config of the line, SQ counters of supremacy-30 and the QFT.  ONE session.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r${ROUND:-06}p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-ladder-base --no-cached-plan --no-configs --no-live-traffic --no-energy"
U="python $R/bench.py --fusion 0 --steps 3 --warmup 1 --no-cpu-baseline"
timeout 1200 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-configs --no-live-traffic --no-energy > $O/bench_driver_style.json 2> $O/bench_driver_style.err
timeout 300 python $R/bench.py --fusion 0 --steps 2 --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err
trace() {  # trace <tag> <skip sweeps> <command...>: kernel trace -> steady-state stats
  local tag=$1 skip=$2; shift 2
  rm -rf /tmp/st_$tag
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -o s -- "$@" > $O/${tag}_trace.log 2>&1
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
trace sup30s1 8 python $R/tools/run_workload.py sup30s1 5  # (round 6: 4 sweeps)
trace sup30s2 8 python $R/tools/run_workload.py sup30s2 5  # (4 sweeps since the level search)
trace qft30c64 6 python $R/tools/run_workload.py qft30c64 5
trace qft33 6 python $R/tools/run_workload.py qft33 4
trace grover34 9 python $R/tools/run_workload.py grover34 2     # the first iteration (9 sweeps) dropped
# the per-gate kernels: dispatches after the warm-up step, per kernel and per target bit of the 30 H gates
rm -rf /tmp/st_unf
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_unf -o s -- $U > $O/unfused_trace.log 2>&1
f=$(find /tmp/st_unf -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python3 - "$f" $O/unfused_kernel_stats.csv <<'PY'
import csv, sys, statistics, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
gate = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_pair', 'k_diag'))]
steady = gate[465:]          # bench.py --fusion 0 --steps 3 --warmup 1: the warm-up step (465 launches) dropped
by = collections.OrderedDict()
for r in steady:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    by.setdefault(name, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
with open(sys.argv[2], 'w') as f:
    f.write('# rocprofv3 --kernel-trace of `bench.py --fusion 0 --steps 3 --warmup 1`: per-gate kernel dispatches after the warm-up step; ms\n')
    f.write('kernel,count,mean_ms,median_ms,min_ms,max_ms,total_ms\n')
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        f.write(f'"{name}",{len(v)},{statistics.mean(v):.4f},{statistics.median(v):.4f},{min(v):.4f},{max(v):.4f},{sum(v):.3f}\n')
    step = steady[:465]
    f.write('# the 30 H gates of one step (dispatch order = gate order): qubit, index bit, kernel, ms, GB/s moved (2 x 16 GiB)\n')
    k = 0
    tot = 0.0
    for i in reversed(range(30)):
        r = step[k]
        ms = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
        tot += ms
        f.write(f'# H qubit {i} bit {29 - i} {r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]} {ms:.4f} {2 * 16 * 2**30 / ms / 1e6:.0f}\n')
        k += 1 + i
    f.write(f'# mean over the 30 H gates: {tot / 30:.4f} ms = {2 * 16 * 2**30 / (tot / 30) / 1e6:.0f} GB/s\n')
    # CU1 launches of that step by the lower of (control, target) index bit
    k = 0
    cls = collections.defaultdict(list)
    for i in reversed(range(30)):
        k += 1
        for j in reversed(range(i)):
            r = step[k]
            lo = min(29 - i, 29 - j)
            cls['bits >= 3' if lo >= 3 else f'lower bit {lo}'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
            k += 1
    for name, v in sorted(cls.items()):
        f.write(f'# CU1 {name}: {len(v)} launches, mean {statistics.mean(v):.4f} ms = {0.5 * 16 * 2**30 / statistics.mean(v) / 1e6:.0f} GB/s algorithmic (S/2 per launch)\n')
print(open(sys.argv[2]).read())
PY
f=$(find /tmp/st_unf -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/unfused_kernel_stats_all_dispatches.csv
pmc pmc_fetch_fused FETCH_SIZE $B
pmc pmc_write_fused WRITE_SIZE $B
python3 $R/tools/collect_traffic.py $O/pmc_fetch_fused.csv $O/pmc_write_fused.csv $O/traffic_fused.json > /dev/null
pmc pmc_fetch_unfused FETCH_SIZE python $R/bench.py --fusion 0 --steps 1 --warmup 0 --no-cpu-baseline
pmc pmc_write_unfused WRITE_SIZE python $R/bench.py --fusion 0 --steps 1 --warmup 0 --no-cpu-baseline
python3 $R/tools/collect_traffic.py $O/pmc_fetch_unfused.csv $O/pmc_write_unfused.csv $O/traffic_unfused.json > /dev/null
for w in sup30 qft30c64 grover34; do
  pmc pmc_fetch_$w FETCH_SIZE python $R/tools/run_workload.py $w 1
  pmc pmc_write_$w WRITE_SIZE python $R/tools/run_workload.py $w 1
  python3 $R/tools/collect_traffic.py $O/pmc_fetch_$w.csv $O/pmc_write_$w.csv $O/traffic_$w.json > /dev/null
  rm -f $O/pmc_fetch_$w.csv $O/pmc_write_$w.csv
done
for w in sup30 qft30; do
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM"; do
    pmc sq_tmp "$set" python $R/tools/run_workload.py $w 1
    echo "## $set" >> $O/sq_counters_$w.txt
    per_sweep $O/sq_tmp.csv >> $O/sq_counters_$w.txt
  done
  rm -f $O/sq_tmp.csv $O/sq_tmp.log
done
rm -f $O/*_trace.log $O/pmc_*.log $O/pmc_fetch_unfused.csv $O/pmc_write_unfused.csv
ls -la $O
head -c 2500 $O/bench_default.json; echo
cat $O/fused_kernel_stats.csv $O/sup30_kernel_stats.csv $O/grover34_kernel_stats.csv $O/unfused_kernel_stats.csv
cat $O/traffic_grover34.json | head -30
