#!/bin/bash
# Infinity-Cache blocking of two dependent sweeps, bytes only (tools/membench/mallbench.hip).
# Output: gpurun_out/r05_mall.txt
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_mall.txt
: > $out
B=tools/membench/mallbench
run() { timeout 60 $B "$@" >> $out 2>&1 || echo "rc=$? for $*" >> $out; }
echo "# baselines: A alone, B alone (bb=21,18), A then B in two launches" >> $out
run 30 21 3 0 0 1 1 1 1
run 30 21 4 0 0 1 1 1 1
run 30 18 4 0 0 1 1 1 1
run 30 21 0 0 0 1 1 1 1
run 30 18 0 0 0 1 1 1 1
run 30 21 0 0 0 0 0 0 0
echo "# mode 1 (same workgroup, block barrier), chip-wide blocks, policies" >> $out
for bb in 18 19 20 21 22; do
  run 30 $bb 1 0 0 1 2 2 1
done
for pol in "1 3 3 1" "1 4 4 1" "1 2 4 1" "1 4 2 1" "0 2 2 0" "0 2 2 1" "1 2 2 0" "1 2 2 4" "1 2 2 2" "1 3 2 1" "1 2 3 1" "1 5 5 1"; do
  run 30 21 1 0 0 $pol
  run 30 18 1 0 0 $pol
done
echo "# unsafe policies (do they even produce the right bytes across XCDs?)" >> $out
run 30 21 1 0 0 1 1 1 1
run 30 21 1 0 0 1 0 0 1
run 30 21 1 0 0 1 0 2 1
run 30 21 1 0 0 1 2 0 1
echo "# mode 1, XCD-local blocks (a block's workgroups share one L2)" >> $out
for bb in 15 16 17 18 19; do
  run 30 $bb 1 0 1 1 0 0 1
  run 30 $bb 1 0 1 1 2 2 1
  run 30 $bb 1 0 1 0 0 0 0
done
echo "# mode 2 (lagged: separate workgroups for B, 'lag' blocks behind A)" >> $out
for lag in 1 2 3 4 6 8; do run 30 21 2 $lag 0 1 2 2 1; done
for lag in 8 16 24 32 48 64; do run 30 18 2 $lag 0 1 2 2 1; done
for lag in 2 4; do run 30 21 2 $lag 0 1 4 4 1; run 30 21 2 $lag 0 0 2 2 0; done
echo done >> $out
