#!/usr/bin/env python3
"""Does the step time of the 30-qubit QFT depend on the ALLOCATION (physical placement of the state's buffers) or on the
process?  One process, N handles created and destroyed in turn, each timed over 8 steps (profiles/r03/bimodal.txt)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, native, workloads  # noqa: E402

n = 30
ops, g8 = workloads.qft_stream(range(n)).arrays()
hold = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
  t0 = time.perf_counter()
  st = device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP)
  st.init_basis(5)
  for _ in range(3):
    st.run_stream(ops, g8); st.flush()
  st.sync()
  t_setup = time.perf_counter() - t0
  st.timer_lap()
  for _ in range(8):
    st.run_stream(ops, g8); st.flush(); st.timer_lap()
  laps = st.timer_laps()
  print(f'handle {trial}: setup {t_setup * 1e3:7.1f} ms  step median {np.median(laps):.3f} ms  min {min(laps):.3f}  max {max(laps):.3f}', flush=True)
  if len(sys.argv) > 2 and sys.argv[2] == 'hold' and trial % 2 == 0:
    hold.append(st)         # keep every other handle alive: the next one lands on other physical pages
  else:
    st.close()
