#!/bin/bash
# round 5: the library at HEAD against round 4's (tools/probes/variants/libqcc_hip_r04.so, built from commit 7be3982's csrc),
# same box, fresh processes, interleaved: is a slower bench line the box or the code?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05vsr04; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -k "argmax or readers" 2>&1 | tail -3 > $O/pytest_argmax.txt
for round in 1 2 3; do for v in r04 head; do for w in sup30 qft30c64 qft33 qft30 grover34; do
  echo "## $v $w round $round" >> $O/ab.txt
  if [ $v = r04 ]; then export QCC_HIP_LIB=$R/tools/probes/variants/libqcc_hip_r04.so; else unset QCC_HIP_LIB; fi
  reps=5; [ $w = grover34 ] && reps=2
  [ $w = grover34 ] && [ $round != 1 ] && continue
  QH_SWEEP_TIMING=1 timeout 600 python tools/run_workload.py $w $reps 2>&1 | grep -a "qh sweeps" | tail -3 >> $O/ab.txt
done; done; done
unset QCC_HIP_LIB
python3 - <<'PY' > gpurun_out/r05vsr04/summary.txt
import re, collections, statistics
cur=None; per=collections.defaultdict(list)
for l in open('gpurun_out/r05vsr04/ab.txt'):
    if l.startswith('##'): cur=tuple(l.split()[1:3])
    elif 'qh sweeps' in l:
        per[cur].append([float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])])
for k in sorted(per, key=lambda k:(k[1],k[0])):
    pp=per[k]; tot=[sum(p) for p in pp]
    print('%-5s %-9s n %2d total median %8.3f min %8.3f | per sweep median'%(k[0],k[1],len(pp),statistics.median(tot),min(tot)), [round(statistics.median(x),3) for x in zip(*pp)])
PY
cat $O/pytest_argmax.txt $O/summary.txt
python tools/probes/single_shot_breakdown.py > $O/single_shot.txt 2>&1; tail -25 $O/single_shot.txt
