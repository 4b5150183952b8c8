cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
for p in 1 2 3; do echo "## process $p"; python tools/probes/alloc_lottery.py 6; done > $O/alloc_lottery.txt 2>&1
echo "## process 4 (every other handle kept alive)" >> $O/alloc_lottery.txt; python tools/probes/alloc_lottery.py 6 hold >> $O/alloc_lottery.txt 2>&1
cat $O/alloc_lottery.txt
