"""Pins oracle/xgates_oracle.c against the reference's golden vectors (CPU only).

Golden vectors come from tools/make_golden.py, which ran the reference's own
compiled src/lib/xgates.cc and its Python fallback loops (src/lib/state.py:80-125).
Tolerance: the oracle and the reference perform the same arithmetic in the same
order; the reference is built -ffast-math, so we allow 4 ulp-scale slack (1e-15
absolute on unit-norm states for complex128, 1e-6 for complex64).
"""
import glob
import os

import numpy as np
import pytest

from tests import oracle_lib
from tests.oracle_lib import NO_CTL

TOL128 = 1e-15
TOL64 = 2e-7


def _load(golden_dir, name):
  return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_single_qubit_every_target(oracle, golden_dir):
  g = _load(golden_dir, 'g3_single.npz')
  n = int(g['nbits'])
  for name, gate, out in zip(g['names'], g['gates'], g['outs']):
    psi = g['psi0'].copy()
    oracle.apply1(psi, gate, n, int(str(name).split(':')[1]))
    assert np.max(np.abs(psi - out)) <= TOL128, name


@pytest.mark.parametrize('fname', ['g4_ctl_n6.npz', 'g4_ctl_n9.npz'])
def test_controlled_every_pair(oracle, golden_dir, fname):
  g = _load(golden_dir, fname)
  n = int(g['nbits'])
  for name, gate, out in zip(g['names'], g['gates'], g['outs']):
    _, c, t = str(name).split(':')
    psi = g['psi0'].copy()
    oracle.applyc(psi, gate, n, int(c), int(t))
    assert np.max(np.abs(psi - out)) <= TOL128, name


def test_complex64(oracle, golden_dir):
  g = _load(golden_dir, 'g7_c64.npz')
  n = int(g['nbits'])
  for name, gate, out in zip(g['names'], g['gates'], g['outs']):
    parts = str(name).split(':')
    psi = g['psi0'].copy()
    if len(parts) == 2:
      oracle.apply1(psi, gate, n, int(parts[1]))
    else:
      oracle.applyc(psi, gate, n, int(parts[1]), int(parts[2]))
    assert psi.dtype == np.complex64
    assert np.max(np.abs(psi - out)) <= TOL64, name


def _trace_files(golden_dir):
  return sorted(glob.glob(os.path.join(golden_dir, 'g5_*.npz'))) + [
      os.path.join(golden_dir, 'g1_qft12.npz'), os.path.join(golden_dir, 'py_fallback.npz')]


def test_recorded_traces(oracle, golden_dir):
  files = _trace_files(golden_dir)
  assert len(files) >= 10
  for f in files:
    g = np.load(f)
    psi = g['init'].astype(np.complex128).copy()
    oracle.run_stream(psi, int(g['nbits']), g['ops'], g['gates'])
    err = np.max(np.abs(psi - g['final']))
    assert err <= 5e-14, (os.path.basename(f), err)


def test_negative_controls_are_covered(golden_dir):
  g = _load(golden_dir, 'g5_negctl.npz')
  ctl = g['ops'][:, 0]
  assert ((ctl < 0) & (ctl != NO_CTL)).sum() >= 10  # circuit_test.py:97-102


def test_qft12_config1_analytic(oracle, golden_dir):
  """G1 also equals the closed form: psi[k] = exp(2 pi i bitrev(x) k / N)/sqrt(N)."""
  g = _load(golden_dir, 'g1_qft12.npz')
  n = 12
  x = int(''.join(str(b) for b in g['bits']), 2)
  xr = int(format(x, '012b')[::-1], 2)
  k = np.arange(1 << n)
  want = np.exp(2j * np.pi * ((xr * k) % (1 << n)) / (1 << n)) / np.sqrt(1 << n)
  assert np.max(np.abs(g['final'] - want)) < 1e-13


def test_libq_plumbing_fixture(golden_dir):
  """G2 (libq, float, little-endian) agrees with G1 (xgates, complex128) to 1e-6."""
  g1 = _load(golden_dir, 'g1_qft12.npz')
  g2 = _load(golden_dir, 'g2_libq_qft12.npz')
  n = 12
  dense = np.zeros(1 << n, dtype=np.complex128)
  for s, a in zip(g2['libq_state'], g2['amp']):
    dense[int(format(int(s), '012b')[::-1], 2)] = a
  assert np.max(np.abs(dense - g1['final'])) < 2e-6


def test_qft22_sampled(oracle, golden_dir):
  g = _load(golden_dir, 'g6_qft22.npz')
  n, x = int(g['nbits']), int(g['x'])
  psi = np.zeros(1 << n, dtype=np.complex128)
  psi[x] = 1
  h = np.array([1, 1, 1, -1], dtype=np.complex128) / np.sqrt(2)
  for i in reversed(range(n)):
    oracle.apply1(psi, h, n, i)
    for j in reversed(range(i)):
      import cmath
      oracle.applyc(psi, np.array([1, 0, 0, cmath.exp(1j * np.pi / 2 ** (i - j))]), n, i, j)
  assert np.max(np.abs(psi[g['idx']] - g['amp'])) < 1e-14
  assert abs(np.vdot(psi, psi).real - float(g['norm2'])) < 1e-12


def test_against_live_reference_build(oracle):
  """When oracle/_ref/libxgates.so is present, compare live on random inputs."""
  xg = oracle_lib.load_ref_xgates()
  if xg is None:
    pytest.skip('oracle/_ref/libxgates.so not built')
  rng = np.random.default_rng(3)
  for n in (1, 2, 3, 11):
    psi0 = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi0 /= np.linalg.norm(psi0)
    for _ in range(12):
      gate = rng.standard_normal(4) + 1j * rng.standard_normal(4)
      t = int(rng.integers(n))
      a, b = psi0.copy(), psi0.copy()
      xg.apply1(a, gate, n, t, 128)
      oracle.apply1(b, gate, n, t)
      assert np.max(np.abs(a - b)) <= 4e-15
      if n > 1:
        c = int((t + 1 + rng.integers(n - 1)) % n)
        a, b = psi0.copy(), psi0.copy()
        xg.applyc(a, gate, n, c, t, 128)
        oracle.applyc(b, gate, n, c, t)
        assert np.max(np.abs(a - b)) <= 4e-15


def test_bad_qubits_are_errors_not_ub(oracle):
  psi = np.zeros(8, dtype=np.complex128)
  with pytest.raises(ValueError):
    oracle.apply1(psi, np.eye(2).reshape(4), 3, 3)
  with pytest.raises(ValueError):
    oracle.apply1(psi, np.eye(2).reshape(4), 3, -1)
  with pytest.raises(ValueError):
    oracle.applyc(psi, np.eye(2).reshape(4), 3, 3, 0)  # positive out-of-range ctl
