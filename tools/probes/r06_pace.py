#!/usr/bin/env python3
"""Round 6: does pacing the waves (QH_SWEEP_PACE: a wave idles pace x 128 cycles before its tile) keep a QFT loop below the socket's
power limit and so avoid the clamp (profiles/r05/qft30_step_cycle.txt: sweeps at 5.39 ms until the limit bites, then 5.98 / 6.25)?
Per pace value: 60 QFT-30 steps in a fresh process, per-step HIP-event times.  GPU box."""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from qcc_amd import device, native, workloads
name = sys.argv[1]
n, bw, steps = 30, 128, 60
if name.startswith('sup30'):
  ops, g8 = workloads.supremacy_stream(30, 20, seed=int(name[6:] or 0) if name[5:6] == 's' else 0).arrays()
elif name == 'grover34':
  n, steps = 34, 6
  ops, g8 = workloads.grover_stream(17, [1, 0] * 8 + [1], iterations=1).arrays()
elif name == 'qft30c64':
  bw = 64
  ops, g8 = workloads.qft_stream(range(n)).arrays()
else:
  n = int(name[3:])
  steps = 60 if n <= 30 else 12
  ops, g8 = workloads.qft_stream(range(n)).arrays()
with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
  st.init_basis(workloads.grover_initial_index(17) if name == 'grover34' else 5)
  for _ in range(3 if n > 30 else 6):
    st.run_stream(ops, g8); st.flush()
  st.sync()
  st.timer_begin(); st.timer_lap()
  for _ in range(steps):
    st.run_stream(ops, g8); st.flush(); st.timer_lap()
  st.timer_end()
  laps = np.array(st.timer_laps())
print('%%s pace %%s: mean %%.3f median %%.3f min %%.3f max %%.3f p10 %%.3f p90 %%.3f ms per step' %% (name, sys.argv[2], laps.mean(), np.median(laps), laps.min(), laps.max(),
      np.percentile(laps, 10), np.percentile(laps, 90)))
''' % R
names = sys.argv[1].split(',') if len(sys.argv) > 1 else ['qft30', 'sup30']
paces = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1, 2, 4, 8, 16, 32]
for rnd in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
  for name in names:
    for pace in paces:
      e = dict(os.environ, QH_PLAN_CACHE='1')
      if pace >= 0:
        e['QH_SWEEP_PACE'] = str(pace)          # (-1: the engine's own rule, sweep_pace())
      r = subprocess.run([sys.executable, '-c', CODE, name, str(pace)], env=e, capture_output=True, text=True, timeout=300)
      print(r.stdout.strip() or r.stderr[-300:], flush=True)
