#!/bin/bash
# round 5: Grover-34 per sweep against the block-index rotation of its launches (contiguous / split-lane tiles)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05grover2; mkdir -p $O
cd $R
for v in "3 6" "0 0" "3 0" "3 3" "0 6" "6 6" "3 9" "3 12"; do
  set -- $v
  echo "## rot contiguous $1 split $2" >> $O/ab.txt
  QH_ROT_EXP_C=$1 QH_ROT_EXP_S=$2 QH_SWEEP_TIMING=1 timeout 600 python tools/run_workload.py grover34 2 2>&1 | grep -a "qh sweeps" | tail -1 >> $O/ab.txt
done
for v in "3 6" "3 0" "3 3" "3 9"; do
  set -- $v
  for w in qft33 qft30 sup30; do
  echo "## $w rot contiguous $1 split $2" >> $O/ab.txt
  QH_ROT_EXP_C=$1 QH_ROT_EXP_S=$2 QH_SWEEP_TIMING=1 timeout 600 python tools/run_workload.py $w 3 2>&1 | grep -a "qh sweeps" | tail -2 >> $O/ab.txt
  done
done
cat $O/ab.txt
