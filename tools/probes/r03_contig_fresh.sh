# Fresh processes, one handle each (what bench.py sees): QH_ALLOC_CONTIG=1 vs 0, interleaved, 12 each.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
{
for p in 1 2 3 4 5 6 7 8 9 10 11 12; do
  for c in 1 0; do echo -n "contig=$c "; QH_ALLOC_CONTIG=$c timeout 300 python tools/probes/alloc_lottery.py 1 2>&1 | tail -1; done
done
} > $O/alloc_contig_fresh.txt 2>&1
cat $O/alloc_contig_fresh.txt
