#!/bin/bash
# SQ counters of the fused bench's sweep kernels (run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU --output-format csv -d /tmp/sq -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/sq.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/sq/**/t_counter_collection.csv', recursive=True)
if not f:
    print(open('/tmp/sq.log').read()[-1500:]); raise SystemExit
rows = [r for r in csv.DictReader(open(f[0])) if 'k_sweep' in r['Kernel_Name']]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
for d, c in list(by.items())[-5:]:
    wc = c.get('SQ_WAVE_CYCLES', 1)
    print(d, {k: (round(v / wc, 3) if k.startswith('SQ_W') or k.startswith('SQ_A') else int(v)) for k, v in c.items()}, 'wave_cycles', int(wc))
PY
