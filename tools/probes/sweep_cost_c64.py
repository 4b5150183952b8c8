#!/usr/bin/env python3
"""complex64: cost of single sweeps by op class on a dense 30-qubit state (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, gates, native, workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bw = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NO = workloads.NO_CTL
PREP = workloads.qft_stream(range(n)).arrays()


def stream(gl):
  ops = np.array([(c, t) for c, t, _ in gl], dtype=np.int32)
  g8 = np.array([np.asarray(g, dtype=np.complex128).reshape(4) for _, _, g in gl]).view(np.float64).reshape(-1, 8)
  return ops, g8


def q(bit):
  return n - 1 - bit


def timed(name, gl, reps=4):
  ops, g8 = stream(gl)
  with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(0x2CB9A5E3 & ((1 << n) - 1))
    st.run_stream(*PREP); st.flush()
    st.run_stream(ops, g8); st.flush(); st.sync()
    st.reset_stats()
    st.timer_begin()
    for _ in range(reps):
      st.run_stream(ops, g8); st.flush()
    ms = st.timer_end() / reps
    s = st.stats()
  k = s['kernels_launched'] // reps
  print(f'{name:50s} {ms:7.3f} ms  {k} sweeps  {s["bytes_swept"] / reps / ms / 1e6:7.0f} GB/s')


h = gates.hadamard()
os.environ['QH_RELAYOUT'] = '0'
timed('one phase group (T on bit 5)', [(NO, q(5), gates.tgate())])
timed('6 H on bits 6..11', [(NO, q(b), h) for b in range(6, 12)])
timed('6 H on lane bits 0..5', [(NO, q(b), h) for b in range(6)])
timed('12 H on bits 0..11', [(NO, q(b), h) for b in range(12)])
qops, qg = workloads.qft_stream(range(n - 12, n)).arrays()
gl = [(int(c), int(t), qg[k].view(np.complex128)) for k, (c, t) in enumerate(qops)]
timed('QFT on bits 0..11 (12 H + 66 CU1)', gl)
timed('9 H on bits 12..20 (in place)', [(NO, q(b), h) for b in range(12, 21)])
del os.environ['QH_RELAYOUT']
timed('full QFT, relayout', [(int(c), int(t), PREP[1][k].view(np.complex128)) for k, (c, t) in enumerate(PREP[0])])
