#!/bin/bash
# Instruction-fetch counters per sweep launch (is the 100+ KiB op interpreter missing the 64 KiB I-cache?)
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-sup30}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "\b\(SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_ICACHE[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_ANY\|SQ_WAIT_INST_ANY\|SQ_INSTS_BRANCH\|SQ_INSTS_CBRANCH[A-Z_]*\|SQC_DCACHE[A-Z_]*\)\b" | sort -u | tr '\n' ' '; echo
for SET in "SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN SQC_DCACHE_REQ SQC_DCACHE_MISSES"; do
echo "## $SET"
rm -rf /tmp/sqw && timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/sqw -o t -- python $R/tools/run_workload.py $W 0 > /tmp/sqw.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/sqw/**/t_counter_collection.csv', recursive=True)
if not f:
    print(open('/tmp/sqw.log').read()[-800:]); raise SystemExit
rows = [r for r in csv.DictReader(open(f[0])) if 'k_sweep' in r['Kernel_Name']]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
for d, c in list(by.items())[:6]:
    print(d, {k: int(v) for k, v in c.items()})
PY
done
