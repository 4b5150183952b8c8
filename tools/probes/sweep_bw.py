#!/usr/bin/env python3
"""Per-gate-kernel HBM bandwidth by bit position (run on the GPU box).

For each target (and a few control) bit positions, launches the single-gate
kernel REPS times between HIP events and prints algorithmic GB/s
(bytes_algorithmic / time).  Used to find bit positions where the streaming
kernels fall off the HBM roofline."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, gates, native  # noqa: E402


def timed(st, fn, reps):
  st.sync()
  st.reset_stats()
  st.timer_begin()
  for _ in range(reps):
    fn()
  ms = st.timer_end()
  s = st.stats()
  return ms / reps, s['bytes_algorithmic'] / reps / (ms / reps * 1e-3) / 1e9


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--qubits', type=int, default=30)
  ap.add_argument('--reps', type=int, default=5)
  ap.add_argument('--fusion', type=int, default=0)
  args = ap.parse_args()
  n = args.qubits
  st = device.DeviceState(n, 128, fusion=args.fusion)
  st.init_basis(0)
  h = gates.hadamard()
  for q in range(n):  # make the state dense (uniform superposition)
    st.apply1(h, q)
  st.sync()
  res = {'qubits': n, 'pair': {}, 'diag1': {}, 'cu1': {}, 'cpair': {}}
  ms, gbps = timed(st, lambda: st.scale(1.0 + 0.0j), args.reps)
  res['scale_inplace'] = {'ms': ms, 'GBps': gbps if gbps else (2 * 16 * 2 ** n) / (ms * 1e-3) / 1e9}
  for bit in list(range(0, 12)) + list(range(12, n, 3)) + [n - 1]:
    q = n - 1 - bit
    ms, gbps = timed(st, lambda: st.apply1(h, q), args.reps)
    res['pair'][bit] = {'ms': round(ms, 4), 'GBps': round(gbps, 1)}
    ms, gbps = timed(st, lambda: st.apply1(gates.tgate(), q), args.reps)
    res['diag1'][bit] = {'ms': round(ms, 4), 'GBps': round(gbps, 1)}
  for cb, tb in [(0, 1), (0, 5), (0, 20), (1, 2), (2, 3), (3, 4), (3, 20), (5, 6), (6, 7), (6, 25), (10, 11),
                 (10, 29), (20, 21), (28, 29), (29, 0), (20, 3)]:
    if cb >= n or tb >= n:
      continue
    ms, gbps = timed(st, lambda: st.applyc(gates.u1(0.3), n - 1 - cb, n - 1 - tb), args.reps)
    res['cu1'][f'{cb},{tb}'] = {'ms': round(ms, 4), 'GBps': round(gbps, 1)}
    ms, gbps = timed(st, lambda: st.applyc(h, n - 1 - cb, n - 1 - tb), args.reps)
    res['cpair'][f'{cb},{tb}'] = {'ms': round(ms, 4), 'GBps': round(gbps, 1)}
  print(json.dumps(res, indent=1))


if __name__ == '__main__':
  main()
