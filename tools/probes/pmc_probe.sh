#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/sqp -o t -- python $R/tools/probes/sweep_probe.py 30 > /tmp/sqp.log 2>&1
grep case /tmp/sqp.log | cut -c1-90
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/sqp/**/t_counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_sweep' in r['Kernel_Name']]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
ids = sorted(by)
# the probe runs: 5 warm sweeps (one QFT), then for each case 3 reps x (sweeps)
for d in ids[5:]:
    c = by[d]
    print(d, 'VALU/tile', round(c['SQ_INSTS_VALU']/524288), 'SALU/tile', round(c['SQ_INSTS_SALU']/524288), 'LDS/tile', round(c['SQ_INSTS_LDS']/524288), 'SMEM/tile', round(c['SQ_INSTS_SMEM']/524288))
PY
