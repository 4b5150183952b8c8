#!/bin/bash
# round 4, GPU session M: three micro-benchmarks that decide the next kernel step
#  (1) lanemap: may the amplitudes of a 128-byte line sit on other lanes than 0-7?  (2) ldsxbench: cost of a lane <-> register
#  exchange through LDS beside busy VALUs  (3) bigtile: one big tile per CU (two-sweep QFT)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04m; mkdir -p $O
cd $R/tools/membench
V="012345:o 012345:i 013452:o 013452:i 045123:o 045123:i 345012:o 345012:i 015234:o 015234:i 023145:o"
( timeout 300 ./lanemap 30 12 20 $V; timeout 300 ./lanemap 30 16 6 $V ) > $O/lanemap.txt 2>&1
timeout 300 ./ldsxbench 2000 > $O/ldsxbench.txt 2>&1
timeout 300 ./bigtile 30 > $O/bigtile.txt 2>&1
cat $O/lanemap.txt $O/ldsxbench.txt $O/bigtile.txt
