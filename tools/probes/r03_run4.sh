cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
timeout 900 python -m pytest tests/test_gpu_exchange.py -m gpu -q -s -k "shard_size" 2>&1 | grep -a "exchange @\|passed\|failed" > gpurun_out/r03d/pytest.log
for s in 1 0; do
  echo "## sup30 QH_PLAN_SEARCH=$s" >> gpurun_out/r03d/sup30.txt
  QH_PLAN_SEARCH=$s QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py sup30 4 2>&1 | grep -a "qh sweeps\|sweeps" >> gpurun_out/r03d/sup30.txt
done
timeout 300 python bench.py --no-cpu-baseline --no-ladder-base > gpurun_out/r03d/bench.json 2> gpurun_out/r03d/bench.err
cat gpurun_out/r03d/pytest.log gpurun_out/r03d/sup30.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03d/bench.json'))
print({k:d[k] for k in ('ms_per_step','median_ms_per_step')}, d['roofline']['frac'], d.get('single_shot_ms'))
print({k:(v.get('ms_per_step'),v.get('median_ms_per_step'),v.get('sweeps_per_step'),v.get('roofline',{}).get('frac')) for k,v in d['configs'].items()})
PY
