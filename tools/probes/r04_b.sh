#!/bin/bash
# round 4, GPU session B: the whole GPU suite on the round's engine changes (exchange watchdog / geometry records, sharded
# layer with one transport family, per-gate kernel shapes), then the per-gate kernels A/B (QH_GATE_SHAPE=0: round-3 shapes)
# and their rocprofv3 kernel trace (steady-state dispatches) -> gpurun_out/r04b/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for round in 1 2; do for s in 0 1; do
  QH_GATE_SHAPE=$s timeout 300 python bench.py --fusion 0 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_unfused_shape${s}_r$round.json 2> $O/bench_unfused_shape${s}_r$round.err
done; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_unf
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_unf -o s -- python $R/bench.py --fusion 0 --steps 3 --warmup 1 --no-cpu-baseline > $O/unfused_trace.log 2>&1
f=$(find /tmp/st_unf -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python3 - "$f" $O/unfused_kernel_stats.csv <<'PY'
import csv, sys, statistics, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
gate = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_pair', 'k_diag'))]
# bench.py --fusion 0 --steps 3 --warmup 1: 4 x 465 launches in the timed loop, then the three class passes (378 + 57 + 30);
# the first 465 (warm-up step) are dropped
steady = gate[465:]
by = collections.OrderedDict()
for r in steady:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    by.setdefault(name, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
with open(sys.argv[2], 'w') as f:
    f.write('# rocprofv3 --kernel-trace of `bench.py --fusion 0 --steps 3 --warmup 1`: per-gate kernel dispatches after the warm-up step; ms\n')
    f.write('kernel,count,mean_ms,median_ms,min_ms,max_ms,total_ms\n')
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        f.write(f'"{name}",{len(v)},{statistics.mean(v):.4f},{statistics.median(v):.4f},{min(v):.4f},{max(v):.4f},{sum(v):.3f}\n')
    # the 30 H gates of ONE step by target bit (dispatch order = gate order: H on qubit i is launch #(sum of earlier gates))
    step = steady[:465]
    f.write('# H gates of one step (k_pair*): index bit, kernel, ms, GB/s moved (2 x 16 GiB)\n')
    k = 0
    for i in reversed(range(30)):
        r = step[k]
        ms = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
        f.write(f'# H qubit {i} bit {29 - i} {r["Kernel_Name"].split("(")[0][:60]} {ms:.4f} {2 * 16 * 2**30 / ms / 1e6:.0f}\n')
        k += 1 + i
print(open(sys.argv[2]).read())
PY
f=$(find /tmp/st_unf -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/unfused_kernel_stats_all_dispatches.csv
for s in 0 1; do for r in 1 2; do python3 -c "
import json,sys
d=json.load(open('$O/bench_unfused_shape${s}_r$r.json'))
print('shape',$s,'round',$r,'ms/step',round(d['ms_per_step'],1), {k.split(' (')[0]+k.split('(')[1][:22]:(round(v['avg_ms'],3),round(v['GBps_algorithmic']),round(v['GBps_moved'])) for k,v in d['roofline']['classes'].items()})
"; done; done
