#!/bin/bash
# round 4, GPU session G: placement probe on / off over fresh processes (QH_PLACEMENT_PROBE), per-step times of the 30-qubit QFT
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
for round in 1 2 3 4 5 6 7 8; do for p in 0 1; do
  echo "## probe=$p round $round" >> $O/probe.txt
  QH_PLACEMENT_PROBE=$p QH_ALLOC_DEBUG=1 timeout 300 python tools/run_workload.py qft30 6 2>&1 | grep -a "placement probe\|step ms" >> $O/probe.txt
done; done
for round in 1 2 3; do for p in 0 1; do
  echo "## probe=$p c64 round $round" >> $O/probe.txt
  QH_PLACEMENT_PROBE=$p QH_ALLOC_DEBUG=1 timeout 300 python tools/run_workload.py qft30c64 6 2>&1 | grep -a "placement probe\|step ms" >> $O/probe.txt
  echo "## probe=$p sup30 round $round" >> $O/probe.txt
  QH_PLACEMENT_PROBE=$p QH_ALLOC_DEBUG=1 timeout 300 python tools/run_workload.py sup30 4 2>&1 | grep -a "placement probe\|step ms" >> $O/probe.txt
done; done
cat $O/probe.txt
