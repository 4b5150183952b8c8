cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 2000 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -120 > gpurun_out/r03c/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-ladder-base > gpurun_out/r03c/bench.json 2> gpurun_out/r03c/bench.err
grep -a "exchange @\|supremacy-28\|passed\|failed\|FAILED\|Error" gpurun_out/r03c/pytest.log | head -40
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03c/bench.json'))
print({k:d[k] for k in ('ms_per_step','median_ms_per_step')}, d['roofline']['frac'], d.get('single_shot_ms'))
print({k:(v.get('ms_per_step'),v.get('sweeps_per_step'),v.get('roofline',{}).get('frac')) for k,v in d['configs'].items()})
PY
