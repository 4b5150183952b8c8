#!/bin/bash
# round 6: what the candidates of plan_best's portfolio really cost.  For supremacy-30 seeds 0..3: every task of the search that
# produced a plan (QH_PLAN_SEARCH_LOG lists them with their predicted time) is forced with QH_PLAN_SEARCH_PICK and run 14 steps
# (plan cache on: the last steps are GPU time only); predicted vs measured -> how good is plan_predicted_ms at ranking tilings?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06cand; mkdir -p $O
cd $R
: > $O/cand.txt
for seed in ${SEEDS:-0 1 2 3}; do
  QH_PLAN_SEARCH_LOG=1 python tools/plan_valu_cost.py sup30s$seed 2> $O/log_$seed.txt > /dev/null
  grep "task .*predicted" $O/log_$seed.txt | sort -u | while read -r line; do
    idx=$(echo "$line" | sed -e 's/.*task \([0-9]*\):.*/\1/')
    pred=$(echo "$line" | sed -e 's/.*predicted \([0-9.]*\) ms.*/\1/')
    desc=$(echo "$line" | sed -e 's/.*task [0-9]*: \(wave bits [0-9] K=[0-9] stream [0-9]\).*/\1/')
    steps=$(QH_PLAN_SEARCH_PICK=$idx timeout 120 python tools/run_workload.py sup30s$seed 14 2>&1 | grep "step ms" | sed -e 's/.*step ms//')
    echo "seed $seed task $idx ($desc) predicted $pred : $steps" >> $O/cand.txt
  done
  steps=$(timeout 120 python tools/run_workload.py sup30s$seed 14 2>&1 | grep "step ms" | sed -e 's/.*step ms//')
  echo "seed $seed default : $steps" >> $O/cand.txt
done
python3 - <<'PY' | tee $O/summary.txt
import re, statistics
for ln in open('gpurun_out/r06cand/cand.txt'):
    head, steps = ln.split(' : ')
    v = [float(x) for x in steps.split()]
    if not v: print(head, 'no data'); continue
    print('%-70s last5 median %.2f min %.2f' % (head, statistics.median(v[-5:]), min(v[-5:])))
PY
