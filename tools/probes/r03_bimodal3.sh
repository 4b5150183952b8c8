cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
{
for p in 1 2 3; do
  echo "## process $p QH_ALLOC_CONTIG=1"; QH_ALLOC_DEBUG=1 QH_ALLOC_CONTIG=1 timeout 300 python tools/probes/alloc_lottery.py 5
done
echo "## QH_ALLOC_CONTIG=1, every other handle kept alive"; QH_ALLOC_DEBUG=1 QH_ALLOC_CONTIG=1 timeout 300 python tools/probes/alloc_lottery.py 6 hold
} > $O/alloc_contig_debug.txt 2>&1
cat $O/alloc_contig_debug.txt
