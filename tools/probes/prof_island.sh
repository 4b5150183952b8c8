#!/bin/bash
# Per-op timeline of a wave inside the sweep kernel (s_memtime records of sampled waves).
#   here (build container):  bash tools/probes/prof_island.sh build     -> gpurun_prof/libqcc_hip_prof.so
#   on the GPU box:          bash tools/probes/prof_island.sh run qft30|sup30 [out.txt]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
LIB=$R/tools/probes/libqcc_hip_prof.so
if [ "$1" = build ]; then
  QH_ISLAND_PROF=1 python3 $R/tools/gen_sweep_asm.py && \
  (cd $R && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DQH_PROF -o $LIB qcc_amd/csrc/engine.hip qcc_amd/csrc/libq_facade.cc) && ls -la $LIB
  exit $?
fi
W=${2:-qft30}
OUT=${3:-$R/gpurun_out/prof_$W.txt}
rm -f $OUT
QH_PLAN_CACHE=0 QH_PROF_OUT=$OUT QCC_HIP_LIB=$LIB python3 $R/tools/run_workload.py $W 1 > /dev/null
# keep the last repetition only
python3 - "$OUT" <<'PY'
import sys
txt = open(sys.argv[1]).read().split('sweep 0 ')
print('sweep 0 ' + txt[-1])
PY
