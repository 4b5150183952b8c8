#!/bin/bash
# round 4, GPU session Q: block rotation of split-lane (gather) sweeps with the short prologue, every workload
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04q; mkdir -p $O
cd $R
for round in 1 2 3; do for v in base s3 s4 s5 s6 c4s4 c2s4; do for w in qft30 qft30c64 sup30 qft33 qft31; do
  echo "## $v $w round $round" >> $O/ab.txt
  unset QCC_HIP_LIB QH_SWEEP_ROT QH_SWEEP_ROT_SPLIT
  case $v in
    base) export QCC_HIP_LIB=$R/tools/probes/variants/libqcc_hip_base.so;;
    s3) export QH_SWEEP_ROT_SPLIT=3;;
    s4) export QH_SWEEP_ROT_SPLIT=4;;
    s5) export QH_SWEEP_ROT_SPLIT=5;;
    s6) export QH_SWEEP_ROT_SPLIT=6;;
    c4s4) export QH_SWEEP_ROT=4;;
    c2s4) export QH_SWEEP_ROT=2 QH_SWEEP_ROT_SPLIT=4;;
  esac
  QH_SWEEP_TIMING=1 timeout 400 python tools/run_workload.py $w 5 2>&1 | grep -a "qh sweeps" | tail -4 >> $O/ab.txt
done; done; done
python3 - <<'PY'
import re, collections, statistics
cur=None; per=collections.defaultdict(list)
for l in open('gpurun_out/r04q/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        per[cur].append(v)
for k in sorted(per, key=lambda k:(k[1],k[0])):
    pp=per[k]; n=len(pp[0]); pp=[p for p in pp if len(p)==n]
    tot=[sum(p) for p in pp]
    print('%-6s %-9s n %2d total median %8.3f mean %8.3f min %8.3f max %8.3f | per sweep median'%(k[0],k[1],len(pp),statistics.median(tot),statistics.mean(tot),min(tot),max(tot)), [round(statistics.median(x),3) for x in zip(*pp)])
PY
