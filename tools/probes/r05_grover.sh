#!/bin/bash
# round 5: Grover-34 (in place, 256 GiB) per sweep, with and without lane seats; the 33-qubit QFT likewise
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05grover; mkdir -p $O
cd $R
for round in 1 2; do for v in off on; do for w in grover34 qft33 qft30; do
  echo "## $v $w round $round" >> $O/ab.txt
  if [ $v = off ]; then export QH_SEATS=0; else unset QH_SEATS; fi
  QH_SWEEP_TIMING=1 timeout 600 python tools/run_workload.py $w 3 2>&1 | grep -a "qh sweeps" | tail -2 >> $O/ab.txt
done; done; done
cat $O/ab.txt
