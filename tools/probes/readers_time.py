#!/usr/bin/env python3
"""Time of the device-side readers (qh_norm2, qh_argmax, qh_prob_bit) on a dense 30-qubit state: host wall clock
around the call (kernel + 8..16 bytes of D2H + stream sync), median of 7."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, native, workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
  st.init_basis(5)
  st.run_stream(*workloads.qft_stream(range(n)).arrays())
  st.sync()
  S = 16 * 2 ** n
  for name, fn in (('norm2', st.norm2), ('argmax', st.argmax), ('prob_bit(0)', lambda: st.prob_bit(0)),
                   ('prob_bit(n-1)', lambda: st.prob_bit(n - 1))):
    ts = []
    for _ in range(7):
      t0 = time.perf_counter()
      fn()
      ts.append(time.perf_counter() - t0)
    m = float(np.median(ts))
    print(f'{name:14s} {m * 1e3:7.3f} ms  ({S / m / 1e9:6.0f} GB/s if it reads the whole state)')
