#!/bin/bash
# Where should lane butterflies leave the LDS (ds_bpermute) path?  QH_LDS_FLOOR = LDS cycles per tile per CU
# a sweep may keep before lane ops move to DPP / v_permlane swaps (planner.h choose_lane_paths).
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in ${WORKLOADS:-qft30 sup30 qft30c64}; do
for f in ${FLOORS:-5500 4500 3500 2500 1500 0}; do
  echo "== $w QH_LDS_FLOOR=$f $(QH_LDS_FLOOR=$f bash $R/tools/trace_workload.sh $w 2>&1 | tail -1)"
done
done
