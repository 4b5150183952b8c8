#!/bin/bash
# timeline of the fused bench (GPU box): every kernel with start offset, duration and the gap before it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tg -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ladder-base --no-cached-plan "$@" > /tmp/tg.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/tg/**/t_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]))
for f in glob.glob('/tmp/tg/**/t_memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'memcpy ' + r.get('Direction', '')))
rows.sort()
first = [i for i, r in enumerate(rows) if 'k_sweep' in r[2]]
if not first:
    print(open('/tmp/tg.log').read()[-2000:]); raise SystemExit
i0 = first[0]
t0 = rows[i0][0]
prev_end = None
for s, e, name in rows[i0:]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f'{(s - t0) / 1e6:9.3f} ms  dur {(e - s) / 1e6:7.3f} ms  gap {gap:8.1f} us  {name}')
    prev_end = max(prev_end or 0, e)
PY
