#!/bin/bash
# round 5: do the steps of a loop over the 30-qubit QFT differ with the layout each one starts from?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 1 2 3; do
QH_SWEEP_TIMING=1 python - <<'PY' 2>&1 | grep -a "qh sweeps\|perm" | head -40
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np
from qcc_amd import device, native, workloads
n = 30
ops, g8 = workloads.qft_stream(range(n)).arrays()
with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
  st.init_basis(5)
  for step in range(14):
    st.run_stream(ops, g8); st.flush(); st.sync()
    bm = (ctypes.c_int32 * 64)()
    native.check(st.lib.qh_get_bitmap(st.h, bm))
    print('perm', step, list(bm)[:n], file=sys.stderr, flush=True)
PY
echo ---
done
