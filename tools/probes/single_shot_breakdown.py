#!/usr/bin/env python3
"""Where the time of `circuit.qc().qft(reg); maxprob()` goes on a warm process (GPU box): Python gate construction, building
the register on the device, handing the queued gates to the engine, the flush, the reader."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['QH_PLAN_CACHE'] = '0'
import numpy as np  # noqa: E402
from qcc_amd.lib import circuit, tensor  # noqa: E402

tensor.set_tensor_width(128)
n = 30
for rep in range(4):
  qc = circuit.qc('single-shot')
  reg = qc.reg(n, rep + 1)
  t = [time.perf_counter()]
  dev = qc._device_ready(); dev.sync(); t.append(time.perf_counter())        # register built on the device
  qc.qft(reg); t.append(time.perf_counter())                                  # 465 gates queued on the host side
  ops_ = np.array(qc._q_ops, dtype=np.int32).reshape(-1, 2)
  gs = np.array(qc._q_gates, dtype=np.complex128).reshape(-1, 4).view(np.float64).reshape(-1, 8); t.append(time.perf_counter())
  qc._drain(); t.append(time.perf_counter())                                  # conversion again + qh_apply_stream
  dev.flush(); t.append(time.perf_counter())                                  # plan + launches (asynchronous)
  dev.sync(); t.append(time.perf_counter())
  qc.maxprob(); t.append(time.perf_counter())
  d = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
  print('rep %d: build register %.2f | qft() python %.2f | numpy conversion %.2f | drain (conversion + apply_stream) %.2f | flush call %.2f | '
        'wait %.2f | maxprob %.2f ms' % ((rep,) + tuple(d)), flush=True)
  del qc
