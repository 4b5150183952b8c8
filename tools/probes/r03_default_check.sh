cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
{
for p in 1 2 3 4 5 6; do
  QH_ALLOC_DEBUG=1 timeout 300 python bench.py --no-configs --no-ladder-base --no-cpu-baseline --no-cached-plan 2> $O/e.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('default bench: ms', round(d['ms_per_step'],3), 'median', round(d['median_ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))"
  grep "qh alloc" $O/e.txt | head -3
done
} > $O/default_check.txt 2>&1
cat $O/default_check.txt
