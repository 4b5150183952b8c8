#!/usr/bin/env python3
"""How a single sweep's time grows with the NUMBER of ops of one class (GPU box): is a wave's op stream
hidden behind the other waves' memory phases (it is for pure FP64 work: tools/membench/overlap.hip)?"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, gates, native, workloads  # noqa: E402

n = 30
NO = workloads.NO_CTL
os.environ['QH_RELAYOUT'] = '0'


def stream(gl):
  ops = np.array([(c, t) for c, t, _ in gl], dtype=np.int32)
  g8 = np.array([np.asarray(g, dtype=np.complex128).reshape(4) for _, _, g in gl]).view(np.float64).reshape(-1, 8)
  return ops, g8


def q(bit):
  return n - 1 - bit


PREP = workloads.qft_stream(range(n)).arrays()


def timed(name, gl, reps=4):
  ops, g8 = stream(gl)
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(0x2CB9A5E3 & ((1 << n) - 1))
    st.run_stream(*PREP); st.flush()
    st.run_stream(ops, g8)
    p = json.loads(st.plan_json())['sweeps']
    st.flush(); st.sync()
    st.reset_stats()
    st.timer_begin()
    for _ in range(reps):
      st.run_stream(ops, g8); st.flush()
    ms = st.timer_end() / reps
    k = st.stats()['kernels_launched'] // reps
  desc = ' '.join(f"[d{s['dense_ops']} bf{s['butterfly_ops']} dpp{s['dpp_ops']} sw{s['lswap_ops']} g{s['groups']}]" for s in p)
  print(f'{name:44s} {ms:7.3f} ms  {k} sweeps {desc}', flush=True)


G = [gates.hadamard(), gates.vgate(), gates.yroot()]


def cyc(bits, count):
  return [(NO, q(bits[i % len(bits)]), G[(i // len(bits) + i) % 3]) for i in range(count)]


timed('one T (floor)', [(NO, q(5), gates.tgate())])
for cnt in (5, 10, 20, 40, 80):
  timed(f'{cnt} butterflies on reg bits 6..10', cyc([6, 7, 8, 9, 10], cnt))
for cnt in (3, 6, 12, 24):
  timed(f'{cnt} butterflies on lane bits 0..2', cyc([0, 1, 2], cnt))
for cnt in (3, 6, 12, 24):
  timed(f'{cnt} butterflies on lane bits 3..5', cyc([3, 4, 5], cnt))
for cnt in (2, 4, 8, 16):
  timed(f'{cnt} butterflies on wave bit 11', cyc([11], cnt))
# diagonal groups: T gates / CZ on varying bits interleaved with one H so that they cannot all merge
t = gates.tgate()
for cnt in (5, 10, 20, 40):
  gl = []
  for i in range(cnt):
    gl.append((NO, q(6 + i % 5), G[i % 3]))
    gl.append((q(6 + (i + 1) % 5), q(6 + (i + 2) % 5), gates.u1(0.1 * (i + 1))))
  timed(f'{cnt} x (reg butterfly + reg-reg phase)', gl)
for cnt in (5, 10, 20, 40):
  gl = []
  for i in range(cnt):
    gl.append((NO, q(6 + i % 5), G[i % 3]))
    gl.append((q(i % 6), q(6 + (i + 2) % 5), gates.u1(0.1 * (i + 1))))
  timed(f'{cnt} x (reg butterfly + lane-reg phase)', gl)
