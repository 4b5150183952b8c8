#!/bin/bash
# Round-3 evidence, part a (GPU box) -> gpurun_out/r03a/: power / clock traces under the sweeps, and the
# 33-qubit QFT (the N=1 point of config 5's ladder) per sweep: times, translation / DRAM-credit / SQ counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocm-smi --showclocks --showpower > $O/smi_idle.txt 2>&1
for w in qft30 sup30; do
  timeout 120 python $R/tools/probes/smi_trace.py $w 25 $O/smi_trace_$w.csv > $O/smi_trace_$w.log 2>&1
done
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
for mode in 1 0; do
  echo "## QH_RELAYOUT=$mode per-sweep ms (QH_SWEEP_TIMING=1)" >> $O/qft33_counters.txt
  QH_RELAYOUT=$mode QH_SWEEP_TIMING=1 timeout 600 python $R/tools/run_workload.py qft33 3 2>&1 | grep -a "qh sweeps" >> $O/qft33_counters.txt
  QH_RELAYOUT=$mode QH_PLAN_VERBOSE=0 python - >> $O/qft33_counters.txt <<PY
import sys; sys.path.insert(0, '$R')
from tests.test_planner_cpu import _plan
from qcc_amd import workloads
ops, g8 = workloads.qft_stream(range(33)).arrays()
for s in _plan(33, ops, g8)['sweeps']:
  print({k: s[k] for k in ('gates', 'dense_ops', 'diag_ops', 'groups', 'regpos', 'lanehi', 'wavepos', 'relayout')})
PY
  for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
             "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
    QH_RELAYOUT=$mode pmc q33_tmp "$set" python $R/tools/run_workload.py qft33 1
    echo "## QH_RELAYOUT=$mode  $set" >> $O/qft33_counters.txt
    per_sweep $O/q33_tmp.csv >> $O/qft33_counters.txt
  done
done
rm -f $O/q33_tmp.csv $O/q33_tmp.log
ls -la $O; cat $O/qft33_counters.txt | head -80; tail -3 $O/smi_trace_qft30.csv; cat $O/smi_trace_qft30.log
