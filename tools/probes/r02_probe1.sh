#!/bin/bash
# Round-2 probe 1 (GPU box): out-of-place tile geometries + the counter list + translation counters
# of the three QFT-30 sweeps.  Writes gpurun_out/r02a/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C=3,4,5,6,7,8,9,10,11
M=12,13,14,15,16,17,18,19,20
H=21,22,23,24,25,26,27,28,29
# split-lane variants as the planner lays sweeps 2/3 out today (lowest three = lane bits)
timeout 300 $R/tools/membench/oopsweep 30 \
  $C:$C:i $M:$M:i $H:$H:i \
  $C:$C $C:$M $C:$H $M:$C $H:$C $M:$M $H:$H $M:$H $H:$M \
  3,4,5,12,13,14,15,16,17:3,4,5,12,13,14,15,16,17:i \
  3,4,5,21,22,23,24,25,26:3,4,5,21,22,23,24,25,26:i \
  $C:3,4,5,12,13,14,15,16,17 $C:3,4,5,21,22,23,24,25,26 \
  $C:18,19,20,12,13,14,15,16,17 $C:27,28,29,21,22,23,24,25,26 \
  > $O/oopsweep.txt 2>&1
cat $O/oopsweep.txt
rocprofv3 -L > $O/counters.txt 2>&1
grep -i -E "utcl|tlb|translat" $O/counters.txt | head -60 > $O/counters_tlb.txt
wc -l $O/counters.txt $O/counters_tlb.txt
head -40 $O/counters_tlb.txt
