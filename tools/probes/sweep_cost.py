#!/usr/bin/env python3
"""Cost of single sweeps by op class (GPU box): one flush = one sweep over a 30-qubit state."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, gates, native, workloads  # noqa: E402

n = 30
NO = workloads.NO_CTL


def stream(gl):
  ops = np.array([(c, t) for c, t, _ in gl], dtype=np.int32)
  g8 = np.array([np.asarray(g, dtype=np.complex128).reshape(4) for _, _, g in gl]).view(np.float64).reshape(-1, 8)
  return ops, g8


def q(bit):
  return n - 1 - bit


def timed(name, gl, reps=4):
  ops, g8 = stream(gl)
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(0x2CB9A5E3 & ((1 << n) - 1))
    st.run_stream(*PREP); st.flush()                 # dense, varied amplitudes (zeros run at a higher clock)
    st.run_stream(ops, g8); st.flush(); st.sync()
    st.reset_stats()
    st.timer_begin()
    for _ in range(reps):
      st.run_stream(ops, g8); st.flush()
    ms = st.timer_end() / reps
    k = st.stats()['kernels_launched'] // reps
  print(f'{name:50s} {ms:7.3f} ms  {k} sweeps')


h = gates.hadamard()
PREP = workloads.qft_stream(range(n)).arrays()
bits = list(range(12))
os.environ['QH_RELAYOUT'] = '0'
timed('identity-ish: T on bit 5 (one diag group)', [(NO, q(5), gates.tgate())])
timed('12 H on bits 0..11', [(NO, q(b), h) for b in bits])
timed('6 H on lane bits 0..5', [(NO, q(b), h) for b in range(6)])
timed('6 H on bits 6..11 (reg + wave)', [(NO, q(b), h) for b in range(6, 12)])
timed('5 H on bits 6..10 (reg only)', [(NO, q(b), h) for b in range(6, 11)])
qops, qg = workloads.qft_stream(range(n - 12, n)).arrays()     # QFT on the 12 lowest index bits
gl = [(int(c), int(t), qg[k].view(np.complex128)) for k, (c, t) in enumerate(qops)]
timed('QFT on bits 0..11 (12 H + 66 CU1)', gl)
timed('66 CU1 on bits 0..11 only', [x for x in gl if x[0] != NO])
def sub(pred):
  return [x for x in gl if x[0] == NO or pred(n - 1 - x[0], n - 1 - x[1])]
timed('12 H + CU1 among lane bits 0..5 (15)', sub(lambda a, b: a < 6 and b < 6))
timed('12 H + CU1 among bits 6..11 (15)', sub(lambda a, b: a >= 6 and b >= 6))
timed('12 H + CU1 lane x reg (36)', sub(lambda a, b: (a < 6) != (b < 6)))
timed('12 H + CU1 with bits 0..2 only', sub(lambda a, b: min(a, b) < 3))
# mid geometry
mid = [0, 1, 2] + list(range(12, 21))
timed('12 H on bits 0,1,2,12..20 (in place)', [(NO, q(b), h) for b in mid])
timed('9 H on bits 12..20 (in place)', [(NO, q(b), h) for b in range(12, 21)])
