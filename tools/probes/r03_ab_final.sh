cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
for round in 1 2 3 4 5 6 7 8 9 10; do for lib in before final; do for w in qft30 qft30c64; do
  echo "## $lib $w round $round" >> $O/ab.txt
  QCC_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/variants/libqcc_$lib.so QH_SWEEP_TIMING=1 timeout 200 python tools/run_workload.py $w 5 2>&1 | grep -a "qh sweeps" | tail -4 >> $O/ab.txt
done; done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list); per=collections.defaultdict(list)
for l in open('gpurun_out/r03n/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    else:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        data[cur].append(sum(v)); per[cur].append(v)
for k in sorted(data, key=lambda k:(k[1],k[0])): print(k, 'median total ms %.3f  min %.3f  n %d'%(statistics.median(data[k]),min(data[k]),len(data[k])), 'per sweep', [round(statistics.median(x),3) for x in zip(*[p for p in per[k] if len(p)==len(per[k][0])])])
PY
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -8
