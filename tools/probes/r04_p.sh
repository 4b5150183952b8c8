#!/bin/bash
# round 4, GPU session P: why do the QFT's gather sweeps scatter (5.4 .. 6.4 ms) with the short prologue?  occupancy (LDS pad),
# staggered wave starts, block rotation -- 30-qubit QFT, fresh processes, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p2; mkdir -p $O
cd $R
for round in 1 2 3 4; do for v in base new pad32 pad40 stag4 stag16 rot4 rot7; do
  echo "## $v qft30 round $round" >> $O/ab.txt
  unset QCC_HIP_LIB QH_SWEEP_LDS_PAD QH_SWEEP_STAGGER QH_SWEEP_ROT
  case $v in
    base) export QCC_HIP_LIB=$R/tools/probes/variants/libqcc_hip_base.so;;
    pad32) export QH_SWEEP_LDS_PAD=8192;;
    pad40) export QH_SWEEP_LDS_PAD=16384;;
    stag4) export QH_SWEEP_STAGGER=4;;
    stag16) export QH_SWEEP_STAGGER=16;;
    rot4) export QH_SWEEP_ROT=4;;
    rot7) export QH_SWEEP_ROT=7;;
  esac
  QH_SWEEP_TIMING=1 timeout 400 python tools/run_workload.py qft30 6 2>&1 | grep -a "qh sweeps" | tail -5 >> $O/ab.txt
done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; per=collections.defaultdict(list)
for l in open('gpurun_out/r04p2/ab.txt'):
    if l.startswith('##'): cur=l.split()[1]
    elif 'qh sweeps' in l:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        if len(v)==3: per[cur].append(v)
for k,pp in per.items():
    tot=[sum(p) for p in pp]
    print('%-7s n %2d total median %.3f mean %.3f min %.3f max %.3f | per sweep median'%(k,len(pp),statistics.median(tot),statistics.mean(tot),min(tot),max(tot)), [round(statistics.median(x),3) for x in zip(*pp)], 'max', [round(max(x),3) for x in zip(*pp)])
PY
