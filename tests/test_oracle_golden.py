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


@pytest.mark.parametrize('fname', ['g3_single.npz', 'g3_single_n10.npz'])
def test_single_qubit_every_target(oracle, golden_dir, fname):
  g = _load(golden_dir, fname)
  n = int(g['nbits'])
  for name, gate, out in zip(g['names'], g['gates'], g['outs']):
    psi = g['psi0'].copy()
    oracle.apply1(psi, gate, n, int(str(name).split(':')[1]))
    assert np.max(np.abs(psi - out)) <= TOL128, name


@pytest.mark.parametrize('fname', [f'g4_ctl_n{n}.npz' for n in (6, 7, 8, 9, 10)])
def test_controlled_every_pair(oracle, golden_dir, fname):
  g = _load(golden_dir, fname)
  n = int(g['nbits'])
  for name, gate, out in zip(g['names'], g['gates'], g['outs']):
    _, c, t = str(name).split(':')
    psi = g['psi0'].copy()
    oracle.applyc(psi, gate, n, int(c), int(t))
    assert np.max(np.abs(psi - out)) <= TOL128, name


@pytest.mark.parametrize('n', [9, 10])
def test_controlled_every_pair_n9_n10(oracle, golden_dir, n):
  """G4 for EVERY ordered (ctl, tgt) at 9 and 10 qubits (tools/make_golden_r3.py: the reference's own applyc; each
  output stored as 16 inner products with stored probe vectors, plus a few outputs in full)."""
  g = _load(golden_dir, 'g4_pairs_n9_n10.npz')
  names = [str(s) for s in g[f'names_n{n}']]
  assert len(names) == 4 * n * (n - 1)
  full = {str(k): v for k, v in zip(g[f'full_names_n{n}'], g[f'full_n{n}'])}
  probes = g[f'probes_n{n}']
  for name, gate, proj in zip(names, g[f'gates_n{n}'], g[f'proj_n{n}']):
    _, c, t = name.split(':')
    psi = g[f'psi0_n{n}'].copy()
    oracle.applyc(psi, gate, n, int(c), int(t))
    assert np.max(np.abs(probes.conj() @ psi - proj)) <= 4e-15, name
    if name in full:
      assert np.max(np.abs(psi - full[name])) <= TOL128, name


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


def test_supremacy20_sampled(oracle, golden_dir):
  """G9: 20-qubit supremacy circuit of the reference (trace + sampled amplitudes)."""
  g = _load(golden_dir, 'g9_supremacy_n20_s0.npz')
  n = int(g['nbits'])
  psi = np.zeros(1 << n, dtype=np.complex128)
  psi[int(g['init_index'])] = 1
  oracle.run_stream(psi, n, g['ops'], g['gates'])
  assert np.max(np.abs(psi[g['idx']] - g['amp'])) < 5e-15
  assert abs(np.vdot(psi, psi).real - float(g['norm2'])) < 1e-12


def libq_cases(golden_dir):
  """Decodes g8_libq_gates.npz: [(width, initval, [(name, a, b, c, gamma)], dense complex64)]."""
  g = _load(golden_dir, 'g8_libq_gates.npz')
  names = [str(x) for x in g['op_names']]
  out, io, off = [], 0, 0
  for w, init, nops in g['case_head']:
    ops = [(names[int(r[0])], int(r[1]), int(r[2]), int(r[3]), float(gm))
           for r, gm in zip(g['case_ops'][io:io + nops], g['case_gamma'][io:io + nops])]
    io += int(nops)
    out.append((int(w), int(init), ops, g['dense'][off:off + (1 << int(w))]))
    off += 1 << int(w)
  return out


def test_reference_libq_gate_set_equals_dense_semantics(oracle, golden_dir):
  """A7 (src/libq/gates.cc:9-151): what the reference's sparse float libq computed for x, y, z, h,
  t, u1, cu1, cx, cz, ccx, walsh on every target / ordered pair equals the dense 2x2 semantics of
  apply1/applyc (libq target t = index bit t = reference qubit width-1-t), to float accuracy.
  This pins the mapping the GPU facade (include/libq.h) implements."""
  import cmath
  s = 1 / np.sqrt(2)
  mats = {'x': [0, 1, 1, 0], 'y': [0, -1j, 1j, 0], 'z': [1, 0, 0, -1], 'h': [s, s, s, -s],
          't': [1, 0, 0, cmath.exp(1j * np.pi / 4)]}
  cases = libq_cases(golden_dir)
  assert len(cases) >= 190
  seen = set()
  for w, init, ops, dense in cases:
    psi = np.zeros(1 << w, dtype=np.complex128)
    psi[int(format(init, f'0{w}b')[::-1], 2)] = 1          # libq basis state -> reference (big-endian) index
    q = lambda t: t                                        # noqa: E731  index bit t of libq == qubit t after the flip
    for name, a, b, c, gamma in ops:
      seen.add(name)
      if name == 'walsh':
        for i in range(a):
          oracle.apply1(psi, np.array(mats['h'], dtype=np.complex128), w, q(i))
      elif name in mats:
        oracle.apply1(psi, np.array(mats[name], dtype=np.complex128), w, q(a))
      elif name == 'u1':
        oracle.apply1(psi, np.array([1, 0, 0, cmath.exp(1j * np.float32(gamma))]), w, q(a))
      elif name == 'cu1':
        oracle.applyc(psi, np.array([1, 0, 0, cmath.exp(1j * np.float32(gamma))]), w, q(a), q(b))
      elif name == 'cx':
        oracle.applyc(psi, np.array(mats['x'], dtype=np.complex128), w, q(a), q(b))
      elif name == 'cz':
        oracle.applyc(psi, np.array(mats['z'], dtype=np.complex128), w, q(a), q(b))
      elif name == 'ccx':                                  # no native doubly-controlled call: by index arithmetic
        idx = np.arange(1 << w)
        bit = lambda i, k: (i >> (w - 1 - k)) & 1          # noqa: E731
        sel = (bit(idx, q(a)) == 1) & (bit(idx, q(b)) == 1)
        psi = np.where(sel, psi[idx ^ (1 << (w - 1 - q(c)))], psi)
      else:
        raise AssertionError(name)
    # reference index (big-endian qubit order) -> libq basis state: bit reversal
    rev = np.array([int(format(k, f'0{w}b')[::-1], 2) for k in range(1 << w)])
    assert np.max(np.abs(psi[rev] - dense)) < 3e-6, (w, init, ops[-1])
  assert seen >= {'x', 'y', 'z', 'h', 't', 'u1', 'cu1', 'cx', 'cz', 'ccx', 'walsh'}


def libq_gate1_prep(w):
  """The libq calls that make the dense entangled register of the libq_gate1 cases (tools/make_golden_r6.py and
  tests/test_gpu_libq_facade.py run exactly these in front of the gate)."""
  return ([('walsh', w, 0, 0, 0.0)] + [('u1', i, 0, 0, 0.21 * (i + 1)) for i in range(w)]
          + [('cu1', 0, w - 2, 0, 0.6), ('h', 2, 0, 0, 0.0), ('cx', 1, w - 1, 0, 0.0)])


def libq_gate1_cases(golden_dir):
  """Decodes g10_libq_gate1.npz: [(width, initval, dense_prep, target, m[4] complex64, dense complex64)]."""
  g = _load(golden_dir, 'g10_libq_gate1.npz')
  out, off = [], 0
  for w, init, dp, t, m in zip(g['width'], g['init'], g['dense_prep'], g['target'], g['m']):
    out.append((int(w), int(init), int(dp), int(t), m, g['dense'][off:off + (1 << int(w))]))
    off += 1 << int(w)
  assert off == g['dense'].size
  return out


def libq_gate1_expected(oracle, w, init, dense_prep, target, m):
  """The dense 2x2 semantics of libq_gate1 through the oracle: libq target t = index bit t = reference qubit t of the
  bit-reversed index; returns the state indexed by libq basis state."""
  import cmath
  s = 1 / np.sqrt(2)
  psi = np.zeros(1 << w, dtype=np.complex128)
  psi[int(format(init, f'0{w}b')[::-1], 2)] = 1
  if dense_prep:
    for name, a, b, _, gamma in libq_gate1_prep(w):
      if name == 'walsh':
        for i in range(a):
          oracle.apply1(psi, np.array([s, s, s, -s], dtype=np.complex128), w, i)
      elif name == 'h':
        oracle.apply1(psi, np.array([s, s, s, -s], dtype=np.complex128), w, a)
      elif name == 'u1':
        oracle.apply1(psi, np.array([1, 0, 0, cmath.exp(1j * np.float32(gamma))]), w, a)
      elif name == 'cu1':
        oracle.applyc(psi, np.array([1, 0, 0, cmath.exp(1j * np.float32(gamma))]), w, a, b)
      elif name == 'cx':
        oracle.applyc(psi, np.array([0, 1, 1, 0], dtype=np.complex128), w, a, b)
      else:
        raise AssertionError(name)
  oracle.apply1(psi, np.asarray(m, dtype=np.complex128), w, target)
  rev = np.array([int(format(k, f'0{w}b')[::-1], 2) for k in range(1 << w)])
  return psi[rev]


def test_reference_libq_gate1_equals_dense_apply1(oracle, golden_dir):
  """A6 (src/libq/libq.h:69, apply.cc:78-176): what the reference's own libq_gate1 computed for unitary and non-unitary
  2x2 matrices on every target of 6-, 8- and 10-qubit registers (dense and single-basis-state inputs) equals apply1's
  dense semantics with m row-major, to float accuracy.  Pins what the facade's libq_gate1 implements."""
  cases = libq_gate1_cases(golden_dir)
  assert len(cases) == 2 * 4 * (6 + 8 + 10)
  assert {(c[0], c[3]) for c in cases} == {(w, t) for w in (6, 8, 10) for t in range(w)}
  for w, init, dp, t, m, dense in cases:
    want = libq_gate1_expected(oracle, w, init, dp, t, m)
    scale = max(1.0, float(np.max(np.abs(want))))
    assert np.max(np.abs(want - dense)) < 3e-6 * scale, (w, init, dp, t)


def test_all_core_variant_equals_serial(golden_dir):
  """oracle_run_stream_c128_mt (the all-core CPU-baseline leg of bench.py) against the serial
  restatement on recorded reference traces and a random stream; serial and OpenMP builds."""
  from qcc_amd import workloads
  for omp in (False, True):
    o = oracle_lib.load(omp=omp)
    for f in ('g5_supremacy_n12_s0.npz', 'g5_grover6.npz', 'g5_qft_iqft_n10.npz'):
      g = _load(golden_dir, f)
      psi = g['init'].astype(np.complex128).copy()
      o.run_stream_mt(psi, int(g['nbits']), g['ops'], g['gates'])
      assert np.max(np.abs(psi - g['final'])) <= 5e-14, (f, omp)
    n = 17
    ops, g8 = workloads.qft_stream(range(n)).arrays()
    a = np.zeros(1 << n, dtype=np.complex128)
    a[12345] = 1
    b = np.empty(1 << n, dtype=np.complex128)
    o.init_basis_mt(b, n, 12345)
    assert np.array_equal(a, b)
    o.run_stream(a, n, ops, g8)
    o.run_stream_mt(b, n, ops, g8)
    assert np.max(np.abs(a - b)) < 1e-15


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
