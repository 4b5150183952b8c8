#!/bin/bash
# round 4: what does a sweep cost when only its op stream runs?  TIMING ONLY (the results are garbage): islands built with
# QH_ISLAND_NOMEM=1 (no tile loads, no stores) and =2 (loads, no stores) against the real ones, per sweep.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for w in sup30 qft30 qft33; do for v in real nomem1 nomem2; do
  echo "## $v $w"
  unset QCC_HIP_LIB; [ $v != real ] && export QCC_HIP_LIB=$R/tools/probes/variants/libqcc_hip_$v.so
  QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 3 2>&1 | grep -a "qh sweeps" | tail -2
done; done
