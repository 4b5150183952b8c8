#!/usr/bin/env python3
"""complex64 30-qubit QFT: per-sweep times (QH_SWEEP_TIMING) for a few steps (GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['QH_SWEEP_TIMING'] = '1'
from qcc_amd import device, native, workloads  # noqa: E402
n = 30
ops, g8 = workloads.qft_stream(range(n)).arrays()
with device.DeviceState(n, 64, fusion=native.QH_FUSE_SWEEP) as st:
  st.init_basis(5)
  for _ in range(4):
    st.run_stream(ops, g8)
    st.flush()
  st.sync()
