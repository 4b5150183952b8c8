#!/bin/bash
# per-launch durations of the fused bench (run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ladder-base --no-cached-plan "$@" > /tmp/tr.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/tr/**/t_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_sweep' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
per = len(d) // 4  # warmup + 3 steps
print('sweeps:', len(d), 'all (ms):', [round(x, 2) for x in d])
print('last step (ms):', [round(x, 2) for x in d[-per:]], 'sum', round(sum(d[-per:]), 2))
PY
