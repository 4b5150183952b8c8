#!/bin/bash
# second pass of the Infinity-Cache probe: the lagged order over block sizes x lag (in MiB), occupancy caps, more policies
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_mall2.txt
: > $out
B=tools/membench/mallbench
run() { timeout 60 $B "$@" >> $out 2>&1 || echo "rc=$? for $*" >> $out; }
echo "# lagged order: block size x lag in MiB (policies nt / sc1 / sc1 / nt)" >> $out
for bb in 15 16 17 18 19 20; do
  blk=$(( (1 << bb) * 16 / 1048576 )); [ $blk -lt 1 ] && blk=0
  for mib in 48 64 80 96 112 128 144 160 192; do
    if [ $bb -eq 15 ]; then lag=$(( mib * 2 )); else lag=$(( mib / blk )); fi
    run 30 $bb 2 $lag 0 1 2 2 1
  done
done
echo "# occupancy caps (dynamic LDS per workgroup): baselines and lagged" >> $out
for lds in 40960 53248 81920; do
  export MALL_LDS=$lds
  run 30 18 0 0 0 1 1 1 1
  for mib in 16 32 48 64 96 128; do run 30 18 2 $(( mib / 4 )) 0 1 2 2 1; done
  for mib in 16 32 48 64 96 128; do run 30 16 2 $mib 0 1 2 2 1; done
done
unset MALL_LDS
echo "# policies on the streams that should NOT stay in the cache (X loads, B stores)" >> $out
for pol in "3 2 2 1" "6 2 2 1" "1 2 2 6" "1 6 6 1" "6 2 2 6" "0 2 2 0" "1 2 2 0" "0 2 2 1" "1 4 4 1" "1 3 3 1"; do
  run 30 18 2 32 0 $pol
  run 30 16 2 112 0 $pol
done
echo done >> $out
