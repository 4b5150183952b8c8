# Both state buffers as halves of ONE physically contiguous allocation, the second at +size+pad: does the step time of
# the 30-qubit QFT depend on the pad (relative placement of the gather stream and the store stream in HBM)?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
{
for rep in 1 2; do
for pad in 0 4 64 1024 2048 2052 4096 32768 1048576 ; do
  echo "## rep $rep QH_ALLOC_PAIR=$pad KiB"; QH_ALLOC_PAIR=$pad timeout 300 python tools/probes/alloc_lottery.py 3 hold 2>&1 | cut -c1-200
done
done
} > $O/alloc_pair.txt 2>&1
cat $O/alloc_pair.txt
