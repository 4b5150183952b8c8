#!/usr/bin/env python3
"""Attribution of the first QFT sweep with wave bits (bits 0..12): subsets of its gates."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, native, workloads  # noqa: E402

n = 30
st = device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP)
st.init_basis(5)
ops, g8 = workloads.qft_stream(range(n)).arrays()
st.run_stream(ops, g8)
st.sync()
isH = ops[:, 0] == workloads.NO_CTL
cb = np.where(isH, -1, n - 1 - ops[:, 0])
tb = n - 1 - ops[:, 1]
lo = np.minimum(np.where(isH, tb, cb), tb)
hi = np.maximum(cb, tb)


def run(name, sel, reps=3):
  st.sync(); st.reset_stats(); st.timer_begin()
  for _ in range(reps):
    st.run_stream(ops[sel], g8[sel]); st.flush()
  ms = st.timer_end() / reps
  s = st.stats()
  print(json.dumps({'case': name, 'gates': int(sel.sum()), 'sweeps': s['sweeps'] // reps, 'ms': round(ms, 3)}))


for top in (10, 11, 12):
  run(f'H 0..{top}', isH & (tb <= top))
  run(f'H 0..{top} + cu1 inside', (isH & (tb <= top)) | (~isH & (hi <= top)))
run('H 11,12 only', isH & (tb >= 11) & (tb <= 12))
run('H 6..12', isH & (tb >= 6) & (tb <= 12))
run('H 6..12 + cu1 inside 6..12', (isH & (tb >= 6) & (tb <= 12)) | (~isH & (lo >= 6) & (hi <= 12)))
H12 = isH & (tb <= 12)
run('H 0..12 + cu1 among 0..10 only', H12 | (~isH & (hi <= 10)))
run('H 0..12 + cu1 touching 11 or 12 only', H12 | (~isH & (hi >= 11) & (hi <= 12)))
run('H 0..12 + cu1 (reg 6..10, 11|12) only', H12 | (~isH & (hi >= 11) & (hi <= 12) & (lo >= 6)))
run('H 0..12 + cu1 (lane 0..5, 11|12) only', H12 | (~isH & (hi >= 11) & (hi <= 12) & (lo <= 5)))
run('H 0..12 + cu1 (11,12) only', H12 | (~isH & (hi == 12) & (lo == 11)))
