#!/usr/bin/env python3
"""Timing of the other BASELINE.json configurations (parity-test cases, not the
bench line): config 3 (30q supremacy depth 20), config 4 (34q Grover, 1 iteration),
and the PCIe-inclusive literal drop-in (qh_host_apply1 on a host buffer)."""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcc_amd import device, gates, native, workloads  # noqa: E402


def timed_stream(n, ops, g8, init, fusion, reps, bw=128):
  with device.DeviceState(n, bw, fusion=fusion) as st:
    st.init_basis(init)
    st.run_stream(ops, g8); st.flush(); st.sync()   # warm-up (also makes the state dense)
    st.reset_stats()
    st.timer_begin()
    for _ in range(reps):
      st.run_stream(ops, g8); st.flush()
    ms = st.timer_end() / reps
    s = st.stats()
  S = (16 if bw == 128 else 8) * 2 ** n
  return {'qubits': n, 'bit_width': bw, 'gates': len(ops), 'fusion': fusion, 'ms': round(ms, 3), 'gate_applies_per_s': round(len(ops) / ms * 1e3, 1),
          'launches': s['kernels_launched'] // reps, 'swept_over_S': round(s['bytes_swept'] / reps / S, 2),
          'algorithmic_over_S': round(s['bytes_algorithmic'] / reps / S, 2),
          'hbm_GBps_swept': round(s['bytes_swept'] / reps / ms / 1e6, 1)}


out = {}
ops, g8 = workloads.supremacy_stream(30, 20, seed=0).arrays()
out['config3_supremacy30_fused'] = timed_stream(30, ops, g8, 0, native.QH_FUSE_SWEEP, 3)
out['config3_supremacy30_unfused'] = timed_stream(30, ops, g8, 0, native.QH_FUSE_OFF, 1)
ops, g8 = workloads.qft_stream(range(30)).arrays()
out['qft30_complex64_fused'] = timed_stream(30, ops, g8, 5, native.QH_FUSE_SWEEP, 3, bw=64)
out['qft31_complex64_fused'] = timed_stream(31, *workloads.qft_stream(range(31)).arrays(), 5, native.QH_FUSE_SWEEP, 3, bw=64)
if '--no34' not in sys.argv:
  nb = 17
  ops, g8 = workloads.grover_stream(nb, [1, 0] * 8 + [1], iterations=1).arrays()
  out['config4_grover34_fused'] = timed_stream(2 * nb, ops, g8, workloads.grover_initial_index(nb), native.QH_FUSE_SWEEP, 1)
# PCIe-inclusive literal drop-in
lib = native.load()
dp = ctypes.POINTER(ctypes.c_double)
for n in (20, 26):
  psi = np.zeros(1 << n, dtype=np.complex128); psi[0] = 1
  g = gates.as8(gates.hadamard())
  lib.qh_host_apply1(psi.ctypes.data, g.ctypes.data_as(dp), n, 0, 128)
  t0 = time.perf_counter()
  reps = 5
  for q in range(reps):
    native.check(lib.qh_host_apply1(psi.ctypes.data, g.ctypes.data_as(dp), n, q, 128))
  dt = (time.perf_counter() - t0) / reps
  out[f'host_dropin_{n}q'] = {'ms_per_gate': round(dt * 1e3, 3), 'pcie_GBps_effective': round(2 * 16 * 2 ** n / dt / 1e9, 2),
                              'gate_applies_per_s': round(1 / dt, 1)}
print(json.dumps(out, indent=1))
