cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
{
for rep in 1 2; do
for padg in 8 16 32 48 64 96 112 128 176 ; do
  echo "## rep $rep QH_ALLOC_PAIR pad $padg GiB"; QH_ALLOC_DEBUG=1 QH_ALLOC_PAIR=$((padg*1048576)) timeout 300 python tools/probes/alloc_lottery.py 2 2>&1 | cut -c1-200
done
done
} > $O/alloc_pair_far.txt 2>&1
cat $O/alloc_pair_far.txt
