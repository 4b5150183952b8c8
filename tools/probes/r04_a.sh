#!/bin/bash
# round 4, GPU session A: (1) launch shapes of the per-gate kernels (tools/membench/pairbench), (2) A/B of the LDS budget of
# lane butterflies (QH_LANE_LDS_BUDGET) on the three sweep-bound workloads, per-sweep times from QH_SWEEP_TIMING.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 600 tools/membench/pairbench 30 6 > $O/pairbench.txt 2>&1
for round in 1 2; do for b in unset 1600 2500 3500 100000; do for w in sup30 qft30 qft33; do
  echo "## $b $w round $round" >> $O/ldsbudget.txt
  if [ $b = unset ]; then
    QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 4 2>&1 | grep -a "qh sweeps\|step ms" | tail -4 >> $O/ldsbudget.txt
  else
    QH_LANE_LDS_BUDGET=$b QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 4 2>&1 | grep -a "qh sweeps\|step ms" | tail -4 >> $O/ldsbudget.txt
  fi
done; done; done
tail -40 $O/pairbench.txt
