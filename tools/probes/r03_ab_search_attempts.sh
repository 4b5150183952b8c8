# Tile search spent on attempts (QH_PLAN_SEARCH_ATTEMPTS=8, default) vs one long walk (=1): supremacy-30 seeds 0..3,
# step times of 5 steps after a warm-up, 3 interleaved rounds; plus the host time of one plan.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O; : > $O/ab.txt
for round in 1 2 3; do for a in 8 1; do for s in 0 1 2 3; do
  echo "## attempts=$a seed $s round $round" >> $O/ab.txt
  QH_PLAN_SEARCH_DEBUG=1 QH_PLAN_SEARCH_ATTEMPTS=$a timeout 200 python tools/run_workload.py sup30s$s 5 2>&1 | grep -a "step ms\|qh search" | sort -u | tail -3 >> $O/ab.txt
done; done; done
cat $O/ab.txt
