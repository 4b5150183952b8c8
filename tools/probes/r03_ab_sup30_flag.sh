cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
for round in 1 2 3 4 5 6; do for v in 0 1; do for w in sup30; do
  echo "## lane3_$v $w round $round" >> $O/ab.txt
  QH_LANE3_LEAST=$v QH_SWEEP_TIMING=1 timeout 200 python tools/run_workload.py $w 5 2>&1 | grep -a "qh sweeps" | tail -4 >> $O/ab.txt
done; done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list); per=collections.defaultdict(list)
for l in open('gpurun_out/r03o/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    else:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        data[cur].append(sum(v)); per[cur].append(v)
for k in sorted(data, key=lambda k:(k[1],k[0])): print(k, 'median total ms %.3f  min %.3f  n %d'%(statistics.median(data[k]),min(data[k]),len(data[k])), 'per sweep', [round(statistics.median(x),3) for x in zip(*[p for p in per[k] if len(p)==len(per[k][0])])])
PY
timeout 400 python tools/fuzz_parity.py 200 701 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | grep -a "passed\|failed" | tail -3
