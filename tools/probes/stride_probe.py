#!/usr/bin/env python3
"""Sweep time vs position of the register bits (H gates only): isolates the memory
access pattern of k_sweep from its arithmetic."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, gates, native
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
st = device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP)
st.init_basis(0)
h = gates.hadamard()
for q in range(n):
  st.apply1(h, q)
st.sync()
def run(bits, reps=3):
  st.sync(); st.reset_stats(); st.timer_begin()
  for _ in range(reps):
    for b in bits:
      st.apply1(h, n - 1 - b)
    st.flush()
  ms = st.timer_end() / reps
  s = st.stats()
  print(json.dumps({'bits': bits, 'ms': round(ms, 3), 'GBps': round(s['bytes_swept'] / reps / ms / 1e6), 'sweeps': s['sweeps'] // reps}))
for bits in ([6,7,8,9,10],[11,12,13,14,15],[16,17,18,19,20],[21,22,23,24,25],[n-5,n-4,n-3,n-2,n-1],[6,7,8,9,n-1],[6,7,8,n-2,n-1],[6,7,n-3,n-2,n-1],[6,n-4,n-3,n-2,n-1],[10,14,18,22,26],[n-1],[20],[12]):
  run(bits)
