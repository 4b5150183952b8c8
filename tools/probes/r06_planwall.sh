#!/bin/bash
# round 6: what planning every step costs a loop (QH_PLAN_CACHE=0, as bench.py runs): supremacy-30 step time against the search budget
# (label changes per task) and the number of streams; the plan cache's GPU-only time beside it.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06pw; mkdir -p $O
cd $R
: > $O/pw.txt
for seed in ${SEEDS:-0 1 2}; do
  for steps in 500000 1000000 2000000 3000000 4000000; do
    for streams in 3 6; do
      s=$(QH_PLAN_CACHE=0 QH_PLAN_SEARCH_STEPS=$steps QH_PLAN_SEARCH_STREAMS=$streams timeout 120 python tools/run_workload.py sup30s$seed 16 2>&1 | grep -E "step ms|'sweeps'" | tr '\n' ' ')
      echo "seed $seed budget $steps streams $streams cache off : $s" >> $O/pw.txt
    done
  done
  s=$(timeout 120 python tools/run_workload.py sup30s$seed 16 2>&1 | grep -E "step ms|'sweeps'" | tr '\n' ' ')
  echo "seed $seed default cache on : $s" >> $O/pw.txt
done
python3 - <<'PY' | tee $O/summary.txt
import re, statistics
for ln in open('gpurun_out/r06pw/pw.txt'):
    head, rest = ln.split(' : ', 1)
    sw = re.search(r"'sweeps': (\d+)", rest)
    v = [float(x) for x in rest.split('step ms')[1].split()]
    print('%-55s sweeps/step %.2f  last 8 steps: median %.2f max %.2f' % (head, int(sw.group(1)) / 17, statistics.median(v[-8:]), max(v[-8:])))
PY
