#!/usr/bin/env python3
"""Marginal cost of the sweep islands' op kinds (GPU box): one flush = ONE sweep over a 30-qubit state with K ops of a kind,
K = 0, 12, 24, 48; the slope is what an op costs a sweep (ms), beside its VALU instruction count priced at full issue rate.
  usage: op_marginal_cost.py [kind ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, gates, native, workloads  # noqa: E402

n = 30
NO = workloads.NO_CTL


def q(bit):
  return n - 1 - bit


def stream(gl):
  ops = np.array([(c, t) for c, t, _ in gl], dtype=np.int32)
  g8 = np.array([np.asarray(g, dtype=np.complex128).reshape(4) for _, _, g in gl]).view(np.float64).reshape(-1, 8)
  return ops, g8


PREP = workloads.qft_stream(range(n)).arrays()


def timed(gl, reps=5):
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
  return ms, k


H, V, Y, T = gates.hadamard(), gates.vgate(), gates.yroot(), gates.tgate()
Z = np.diag([1.0, -1.0]).astype(np.complex128)
REG = [6, 7, 8, 9, 10]          # register bits of the first sweep's tile (contiguous lanes 0..5, wave bit 11)
BASE = [(NO, q(b), H) for b in range(12)]    # fixes the tile: 12 low bits, as the QFT's first sweep


def seq(kind, K):
  gl = list(BASE)
  for k in range(K):
    r = REG[k % 5]
    r2 = REG[(k + 2) % 5]
    if kind == 'reg_bfly':
      gl.append((NO, q(r), [H, V, Y][k % 3]))
    elif kind == 't_then_bfly':
      gl += [(NO, q(r), T), (NO, q(r), [V, Y][k % 2])]
    elif kind == 'cz_regreg_then_bfly':
      gl += [(q(r2), q(r), Z), (NO, q(r), [V, Y][k % 2])]
    elif kind == 'cz_lanereg_then_bfly':
      gl += [(q(k % 3), q(r), Z), (NO, q(r), [V, Y][k % 2])]
    elif kind == 'cz_far_then_bfly':
      gl += [(q(20 + k % 8), q(r), Z), (NO, q(r), [V, Y][k % 2])]
    elif kind == 'dpp_lane0':
      gl.append((NO, q(0), [H, V, Y][k % 3]))
    elif kind == 'dpp_lane2':
      gl.append((NO, q(2), [H, V, Y][k % 3]))
    elif kind == 't_then_dpp_lane1':
      gl += [(NO, q(1), T), (NO, q(1), [V, Y][k % 2])]
    elif kind == 'lane45_alternating':
      gl.append((NO, q(4 + k % 2), [V, Y][k % 2]))
    elif kind == 'wave_bit':
      gl.append((NO, q(11), [V, Y][k % 2]))
      gl.append((NO, q(r), [V, Y][k % 2]))
  return gl


KINDS = ['reg_bfly', 't_then_bfly', 'cz_regreg_then_bfly', 'cz_lanereg_then_bfly', 'cz_far_then_bfly', 'dpp_lane0', 'dpp_lane2',
         't_then_dpp_lane1', 'lane45_alternating', 'wave_bit']
kinds = sys.argv[1:] or KINDS
os.environ['QH_RELAYOUT'] = '0'
base_ms, _ = timed(seq('reg_bfly', 0))
print(f'base sweep (12 H on the 12 low bits): {base_ms:.3f} ms; full-rate price of 64 FP64 instructions per tile: '
      f'{64 * (1 << 19) * 4 / (1024 * 2.05e9) * 1e3:.4f} ms at 2.05 GHz')
for kind in kinds:
  row = []
  for K in (12, 24, 48):
    ms, k = timed(seq(kind, K))
    row.append((K, ms, k))
  slope = (row[-1][1] - row[0][1]) / (row[-1][0] - row[0][0])
  print(f'{kind:24s} ' + '  '.join(f'K={K}: {ms:6.3f} ms ({k} sweep)' for K, ms, k in row) + f'   per op {slope * 1e3:6.1f} us')
