#!/usr/bin/env python3
"""Differential fuzzing of the fused path against the CPU oracle (run on the GPU box).

Random circuits of random shape (qubits, gate mix, control density, multi-controls through
qh_apply_bits, long diagonal runs, butterfly-only stretches) are planned and executed by the
engine and replayed by oracle/xgates_oracle.c; any amplitude differing by more than the
tolerance is a failure.  usage: fuzz_parity.py [SECONDS] [SEED]
The oracle is test infrastructure (see its header); this tool is a test."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcc_amd import device, gates, native  # noqa: E402
from tests import oracle_lib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = oracle_lib.load()
NO = oracle_lib.NO_CTL
argmax_bad = []


def rand_unitary(rng):
  m = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
  q, r = np.linalg.qr(m)
  return q * (np.diag(r) / np.abs(np.diag(r)))


def pools(rng):
  v, yr, h = gates.vgate(), gates.yroot(), gates.hadamard()
  bf = [h, yr, v, np.conj(np.asarray(yr).reshape(2, 2).T), np.conj(np.asarray(v).reshape(2, 2).T)]
  diag = [gates.tgate(), gates.sgate(), gates.pauli_z(), gates.u1(rng.uniform(0, 6)), gates.rz(rng.uniform(0, 6))]
  real = [gates.pauli_x(), gates.ry(rng.uniform(0, 3)) if hasattr(gates, 'ry') else gates.hadamard()]
  gen = [rand_unitary(rng), gates.rx(rng.uniform(0, 3)), gates.pauli_y()]
  if rng.random() < 0.25:   # non-unitary operators the reference also pushes through apply1
    diag += [np.array([[1, 0], [0, 0]]), np.array([[0, 0], [0, 1]]), np.array([[0.5, 0], [0, 2.0]]), np.array([[0, 0], [0, 0]])]
    gen += [np.array([[0, 1], [0, 0]]), np.array([[0, 0], [1, 0]]), np.array([[1, 1], [1, 1]]) * 0.5]
  return bf, diag, real, gen


def one_case(rng, case):
  n = int(rng.integers(7, 22)) if rng.random() < 0.9 else int(rng.integers(22, 25))
  bw = 128 if rng.random() < 0.8 else 64
  if os.environ.get('FUZZ_BW'):
    bw = int(os.environ['FUZZ_BW'])
  ngates = int(rng.integers(20, 500)) if n < 22 else int(rng.integers(20, 120))
  gsh = int(rng.integers(0, 3)) if n >= 10 and rng.random() < 0.3 else 0   # shard bits (top qubits 0..gsh-1)
  w = rng.dirichlet([1, 1, 1, 1])              # butterfly / diagonal / real / general mix
  pctl = rng.uniform(0, 0.8)
  pmulti = rng.uniform(0, 0.3)
  focus = rng.random() < 0.4                    # gates concentrated on few qubits
  hot = rng.choice(n, size=max(2, n // 3), replace=False)
  bf, diag, real, gen = pools(rng)
  dt = np.complex128 if bw == 128 else np.complex64
  psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
  psi = (psi / np.linalg.norm(psi)).astype(dt)
  want = psi.copy()
  stream = []
  for _ in range(ngates):
    kind = rng.choice(4, p=w)
    g = [bf, diag, real, gen][kind]
    g = np.asarray(g[int(rng.integers(len(g)))], dtype=np.complex128).reshape(4)
    t = int(rng.choice(hot)) if focus and rng.random() < 0.8 else int(rng.integers(n))
    if t < gsh and kind != 1:
      t = gsh + t % (n - gsh)                   # dense gates stay on local qubits (the exchange layer is tested elsewhere)
    ctl = []
    if rng.random() < pctl:
      k = 1 + (int(rng.integers(1, 4)) if rng.random() < pmulti else 0)
      others = [q for q in range(n) if q != t]
      ctl = [int(c) for c in rng.choice(others, size=min(k, len(others)), replace=False)]
    stream.append((ctl, t, g))
  # oracle: nested controls = gate applied where all control qubits are 1
  for ctl, t, g in stream:
    gq = g.astype(dt)
    if not ctl:
      orc.apply1(want, gq, n, t)
    elif len(ctl) == 1:
      orc.applyc(want, gq, n, ctl[0], t)
    else:
      idx = np.arange(1 << n, dtype=np.uint64)
      mask = np.ones(1 << n, dtype=bool)
      for c in ctl:
        mask &= ((idx >> np.uint64(n - 1 - c)) & np.uint64(1)).astype(bool)
      tmp = want.copy()
      orc.apply1(tmp, gq, n, t)
      want[mask] = tmp[mask]
  nloc = n - gsh
  got = np.empty_like(psi)
  for shard in range(1 << gsh):                 # every shard of the state on its own handle, one after the other
    with device.DeviceState(nloc, bw, fusion=native.QH_FUSE_SWEEP) as st:
      if gsh:
        st.set_shard(n, shard)
      st.upload(psi[shard << nloc: (shard + 1) << nloc])
      for ctl, t, g in stream:
        cm = 0
        for c in ctl:
          cm |= 1 << (n - 1 - c)
        st.apply_bits(cm, n - 1 - t, g)
      am = st.argmax() if not gsh else None          # (flushes: behind a fused flush the last sweep's per-unit maxima answer it)
      got[shard << nloc: (shard + 1) << nloc] = st.download()
      s = st.stats()
      if am is not None:
        pr = got.real.astype(np.float64) ** 2 + got.imag.astype(np.float64) ** 2
        top = float(pr.max())
        if not (0 <= am[0] < (1 << n) and pr[am[0]] >= top * (1 - 1e-6 if bw == 64 else 1 - 4e-16) - 1e-300 and abs(am[1] - top) <= 1e-6 * top + 1e-300 if bw == 64
                else 0 <= am[0] < (1 << n) and pr[am[0]] >= top * (1 - 4e-16) - 1e-300 and abs(am[1] - top) <= 4e-16 * top + 1e-300):
          print(json.dumps({'FAIL': case, 'what': 'argmax', 'n': n, 'bw': bw, 'got': [int(am[0]), float(am[1])], 'want_p': top,
                            'p_at_got': float(pr[am[0]])}), flush=True)
          argmax_bad.append(case)
  err = float(np.max(np.abs(got - want)))
  tol = 2e-11 if bw == 128 else 2e-4 * max(1.0, ngates / 100)
  ok = err <= tol and not (argmax_bad and argmax_bad[-1] == case)
  if not ok:
    print(json.dumps({'FAIL': case, 'n': n, 'bw': bw, 'gates': ngates, 'err': err, 'sweeps': s['sweeps'], 'env': {k: v for k, v in os.environ.items() if k.startswith('QH_')}}), flush=True)
  return ok, n, s['sweeps']


t0 = time.time()
case = fails = 0
while time.time() - t0 < budget:
  rng = np.random.default_rng(seed0 * 100003 + case)
  ok, n, sw = one_case(rng, case)
  fails += not ok
  case += 1
print(json.dumps({'cases': case, 'failures': fails, 'seconds': round(time.time() - t0, 1), 'seed': seed0,
                  'env': {k: v for k, v in os.environ.items() if k.startswith('QH_')}}))
sys.exit(1 if fails else 0)
