#!/bin/bash
# round 4, GPU session C: the DIAG-group classes of the sweep islands (uniform factors from SGPRs, sign groups without a
# vector prologue, outside-bit sign groups decided on the scalar unit, c initialised lazily): parity + fuzz on the new
# build, then A/B against the build before (tools/probes/variants/) -> gpurun_out/r04c/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_relayout.py tests/test_gpu_lib.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log
tail -4 $O/pytest_subset.log
for s in 11 12 13; do timeout 200 python tools/fuzz_parity.py 45 $s 2>&1 | tail -2 >> $O/fuzz.txt; done
QH_LANE_VALU=2 QH_WAVE_BITS=2 timeout 200 python tools/fuzz_parity.py 45 21 2>&1 | tail -2 >> $O/fuzz.txt
tail -8 $O/fuzz.txt
for round in 1 2 3; do for lib in before classes; do for w in sup30 qft30 qft33 qft30c64; do
  echo "## $lib $w round $round" >> $O/ab.txt
  QCC_HIP_LIB=$R/tools/probes/variants/libqcc_$lib.so QH_SWEEP_TIMING=1 timeout 300 python tools/run_workload.py $w 4 2>&1 | grep -a "qh sweeps\|step ms" | tail -4 >> $O/ab.txt
done; done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; data=collections.defaultdict(list); per=collections.defaultdict(list)
for l in open('gpurun_out/r04c/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        data[cur].append(sum(v)); per[cur].append(v)
for k in sorted(data, key=lambda k:(k[1],k[0])):
    n=len(per[k][0]); pp=[p for p in per[k] if len(p)==n]
    print(k, 'median total ms %.3f  min %.3f  n %d'%(statistics.median(data[k]),min(data[k]),len(data[k])), 'per sweep', [round(statistics.median(x),3) for x in zip(*pp)])
PY
