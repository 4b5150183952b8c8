#!/bin/bash
# GPU box: per-sweep times of the 30-qubit QFT, supremacy-30 and the complex64 QFT for every library variant in
# tools/probes/variants (QCC_HIP_LIB), three rounds interleaved so that drift of the box hits all variants alike.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03v
mkdir -p $O
for round in 1 2 3 4 5 6; do
  for lib in $R/tools/probes/variants/libqcc_*.so; do
    tag=$(basename $lib .so)
    for w in qft30 qft30c64; do
      echo "## $tag $w round $round $EXTRA" >> $O/variants.txt
      QCC_HIP_LIB=$lib QH_SWEEP_TIMING=1 timeout 200 python $R/tools/run_workload.py $w 4 2>&1 | grep -a "qh sweeps" | tail -3 >> $O/variants.txt
    done
  done
done
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list)
for l in open('/root/repo/gpurun_out/r03v/variants.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    else:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        data[cur].append(sum(v))
for k in sorted(data): print(k, 'median total ms %.3f  min %.3f  n %d'%(statistics.median(data[k]),min(data[k]),len(data[k])))
PY
# (QH_RELAYOUT_CONTIG=1 -- contiguous tiles with layout exchanges stored into the second buffer as they are -- was
# measured in batch 1: the QFT's first sweep 6.0-6.1 -> 6.15-6.45 ms; left off)
