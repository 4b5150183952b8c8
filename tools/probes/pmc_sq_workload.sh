#!/bin/bash
# SQ counters per sweep launch of one workload (tools/run_workload.py); run on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-sup30}
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM"; do
rm -rf /tmp/sqw && timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/sqw -o t -- python $R/tools/run_workload.py $W 0 > /tmp/sqw.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/sqw/**/t_counter_collection.csv', recursive=True)
if not f:
    print(open('/tmp/sqw.log').read()[-1500:]); raise SystemExit
rows = [r for r in csv.DictReader(open(f[0])) if 'k_sweep' in r['Kernel_Name']]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
for d, c in by.items():
    print(d, {k: int(v) for k, v in c.items()})
PY
done
