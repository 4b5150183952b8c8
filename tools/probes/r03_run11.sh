cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_relayout.py tests/test_gpu_exchange.py tests/test_gpu_lib.py tests/test_gpu_fullsize.py -m gpu -q -k "not shard_size" 2>&1 | grep -a "passed\|failed\|FAILED\|Error" | tail -8 > $O/pytest.log; cat $O/pytest.log
for round in 1 2 3 4 5 6 7 8 9 10 11 12; do for lib in before ltab_lds_dma; do
  echo "## $lib qft30 round $round" >> $O/ab.txt
  QCC_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/variants/libqcc_$lib.so QH_SWEEP_TIMING=1 timeout 200 python tools/run_workload.py qft30 6 2>&1 | grep -a "qh sweeps" | tail -4 >> $O/ab.txt
done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list); per=collections.defaultdict(list)
for l in open('gpurun_out/r03l/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    else:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        data[cur].append(sum(v)); per[cur].append(v)
for k in sorted(data): print(k, 'median total ms %.3f  mean %.3f min %.3f  n %d'%(statistics.median(data[k]),statistics.mean(data[k]),min(data[k]),len(data[k])), 'per sweep', [round(statistics.median(x),3) for x in zip(*per[k])])
PY
