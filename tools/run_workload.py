#!/usr/bin/env python3
"""Runs one named workload fused, `reps` times (for rocprofv3 traces): qft30 | sup30 | sup30sK (seed K) | grover34 |
qft30c64 | qftNN.  Prints the handle's statistics and the mean step time over the last `reps` steps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcc_amd import device, native, workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'sup30'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bw = 128
if name.startswith('sup30'):
  n, init = 30, 0
  ops, g8 = workloads.supremacy_stream(30, 20, seed=int(name[6:]) if name[5:6] == 's' else 0).arrays()
elif name == 'grover34':
  n, init = 34, workloads.grover_initial_index(17)
  ops, g8 = workloads.grover_stream(17, [1, 0] * 8 + [1], iterations=1).arrays()
elif name == 'qft30c64':
  n, init, bw = 30, 5, 64
  ops, g8 = workloads.qft_stream(range(30)).arrays()
elif name.startswith('qft') and name[3:].isdigit():
  n, init = int(name[3:]), 5
  ops, g8 = workloads.qft_stream(range(n)).arrays()
else:
  n, init = 30, 5
  ops, g8 = workloads.qft_stream(range(30)).arrays()
with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
  st.init_basis(init)
  st.run_stream(ops, g8)
  st.flush()
  st.sync()
  st.timer_lap()
  for _ in range(reps):
    st.run_stream(ops, g8)
    st.flush()
    st.timer_lap()
  laps = st.timer_laps()
  st.sync()
  print(st.stats())
  print(f'{name}: step ms ' + ' '.join(f'{x:.3f}' for x in laps))
