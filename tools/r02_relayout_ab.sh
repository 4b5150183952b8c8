#!/bin/bash
# per-sweep durations of the 30-qubit QFT bench with and without relayout sweeps (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02i
mkdir -p $O
for v in 1 0; do
  echo "== QH_RELAYOUT=$v" >> $O/ab.txt
  QH_RELAYOUT=$v bash $R/tools/trace_sweeps.sh >> $O/ab.txt 2>&1
  QH_RELAYOUT=$v python $R/bench.py --no-cpu-baseline --no-ladder-base --steps 10 --warmup 3 2>&1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],4))" >> $O/ab.txt
done
cat $O/ab.txt
