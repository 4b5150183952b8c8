#!/bin/bash
# round 5: lane / register / wave roles of far tiles at 34 qubits (QH_ROLE_EXP, an experiment switch): Grover-34 per sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05roles; mkdir -p $O
cd $R
for m in 0 1 2 3; do
  echo "## role mode $m" >> $O/ab.txt
  QH_ROLE_EXP=$m QH_SWEEP_TIMING=1 timeout 600 python tools/run_workload.py grover34 2 2>&1 | grep -a "qh sweeps" | tail -1 >> $O/ab.txt
done
for m in 0 1 2 3; do for w in qft33 sup30; do
  echo "## $w role mode $m" >> $O/ab.txt
  QH_ROLE_EXP=$m QH_SWEEP_TIMING=1 timeout 600 python tools/run_workload.py $w 4 2>&1 | grep -a "qh sweeps" | tail -2 >> $O/ab.txt
done; done
cat $O/ab.txt
