#!/bin/bash
# per-launch sweep durations of one workload (tools/run_workload.py), last repetition
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trw && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trw -o t -- python $R/tools/run_workload.py "$1" 2 > /tmp/trw.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/trw/**/t_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_sweep' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
per = len(d) // 3
print('sweeps/rep:', per, 'ms:', [round(x, 2) for x in d[-per:]], 'sum', round(sum(d[-per:]), 2))
PY
