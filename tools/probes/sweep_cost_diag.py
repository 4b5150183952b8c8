#!/usr/bin/env python3
"""Which diagonal groups of a one-sweep circuit cost what (GPU box): 12 H on index bits 0..11 plus subsets of the QFT's
controlled phases among bits 6..11 (register bits 6..10, wave bit 11)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['QH_RELAYOUT'] = '0'
from qcc_amd import device, native, workloads  # noqa: E402

n = 30
NO = workloads.NO_CTL
PREP = workloads.qft_stream(range(n)).arrays()
qops, qg = workloads.qft_stream(range(n - 12, n)).arrays()
gl = [(int(c), int(t), k) for k, (c, t) in enumerate(qops)]


def timed(name, sel, reps=4):
  ops = np.array([(c, t) for c, t, _ in sel], dtype=np.int32)
  g8 = np.array([qg[k] for _, _, k in sel])
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(0x2CB9A5E3 & ((1 << n) - 1))
    st.run_stream(*PREP); st.flush()
    st.run_stream(ops, g8); st.flush(); st.sync()
    st.reset_stats()
    st.timer_begin()
    for _ in range(reps):
      st.run_stream(ops, g8); st.flush()
    ms = st.timer_end() / reps
  print(f'{name:60s} {ms:7.3f} ms  ({len(sel) - 12} CU1)', flush=True)


def sub(pred):
  return [x for x in gl if x[0] == NO or pred(n - 1 - x[0], n - 1 - x[1])]


timed('12 H only', sub(lambda a, b: False))
timed('+ CU1 among bits 6..11 (15)', sub(lambda a, b: a >= 6 and b >= 6))
timed('+ CU1 among register bits 6..10 (10)', sub(lambda a, b: a >= 6 and b >= 6 and max(a, b) <= 10))
timed('+ CU1 between wave bit 11 and 6..10 (5)', sub(lambda a, b: a >= 6 and b >= 6 and max(a, b) == 11))
for lo in range(6, 11):
  timed(f'+ CU1({lo}, 11) alone', sub(lambda a, b: min(a, b) == lo and max(a, b) == 11))
for lo, hi in ((6, 7), (6, 10), (9, 10), (7, 9)):
  timed(f'+ CU1({lo}, {hi}) alone', sub(lambda a, b: min(a, b) == lo and max(a, b) == hi))
timed('+ CU1 among lane bits 0..5 (15)', sub(lambda a, b: a < 6 and b < 6))
timed('+ CU1 lane x reg (36)', sub(lambda a, b: (a < 6) != (b < 6)))
