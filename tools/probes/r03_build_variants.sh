#!/bin/bash
# A/B builds of the engine with other cache-policy bits on the sweep kernel's tile loads / stores
# (tools/gen_sweep_asm.py QH_ISLAND_LD_BITS / QH_ISLAND_ST_BITS) -> tools/probes/variants/libqcc_<tag>.so
# (git-ignored; travels to the GPU box with the snapshot).  Run here, then tools/probes/r03_run_variants.sh there.
R=$(cd "$(dirname "$0")/../.." && pwd)
V=$R/tools/probes/variants
mkdir -p $V
build() {  # build <tag> <load bits> <store bits>
  local tag=$1 ld=$2 st=$3 d=/tmp/qh_var_$1
  rm -rf $d && mkdir -p $d/qcc_amd/csrc $d/include
  cp $R/qcc_amd/csrc/*.h $R/qcc_amd/csrc/*.hip $R/qcc_amd/csrc/*.cc $R/qcc_amd/csrc/sweep_handlers.inc $d/qcc_amd/csrc/
  cp $R/include/*.h $d/include/
  QH_ISLAND_LD_BITS="$ld" QH_ISLAND_ST_BITS="$st" QH_ISLAND_OUT=$d/qcc_amd/csrc python3 $R/tools/gen_sweep_asm.py > /dev/null
  (cd $d && hipcc --offload-arch=gfx950 -O3 -std=c++17 $EXTRA_FLAGS -shared -fPIC -o $V/libqcc_$tag.so qcc_amd/csrc/engine.hip qcc_amd/csrc/libq_facade.cc 2>&1 | grep -v "warning: ignoring\|^$" | head -5)
  echo "built $tag: loads '$ld' stores '$st'"
}
# round-3 batch 3 (QH_ISLAND_BATCH 4 / 8 / 16 slot offsets per scalar round trip): no difference beyond the noise
# round-3 batch 4 (-DQH_SKIP_LTAB_COPY / -DQH_ZERO_LTAB: what the lane-table copy in front of the loads costs): complex64 -6 %
# round-3 batch 5: a window on the tile loads in flight per wave (QH_ISLAND_LOAD_WINDOW)
build nt_nt "nt" "nt" &
QH_ISLAND_LOAD_WINDOW=8 build win8 "nt" "nt" &
QH_ISLAND_LOAD_WINDOW=16 build win16 "nt" "nt" &
wait
ls -la $V
