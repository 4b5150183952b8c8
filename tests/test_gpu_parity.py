"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU
oracle and the reference's golden vectors.

Tolerance (BASELINE.json north_star): max-abs amplitude difference <= 1e-10 for
complex128; we assert 1e-12 on the small cases (the arithmetic is the same
28 flops per pair, differing only by FMA contraction) and 1e-10 at full size.
complex64: 2e-6 (reference default precision).
"""
import ctypes
import glob
import os

import numpy as np
import pytest

from qcc_amd import device, gates, native, workloads
from tests.oracle_lib import NO_CTL

pytestmark = pytest.mark.gpu
TOL = 1e-12
FUSIONS = [native.QH_FUSE_OFF, native.QH_FUSE_SWEEP]


def _rand_state(rng, n, dtype=np.complex128):
  p = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
  return (p / np.linalg.norm(p)).astype(dtype)


def _rand_unitary(rng):
  m = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
  q, r = np.linalg.qr(m)
  return q * (np.diag(r) / np.abs(np.diag(r)))


def _gate_pool(rng):
  return [gates.hadamard(), gates.pauli_x(), gates.pauli_y(), gates.pauli_z(), gates.sgate(),
          gates.tgate(), gates.vgate(), gates.yroot(), gates.u1(0.37), gates.rz(0.9),
          gates.rx(0.4), _rand_unitary(rng), _rand_unitary(rng)]


@pytest.mark.parametrize('fusion', FUSIONS)
def test_golden_single_and_controlled(golden_dir, fusion):
  for fname in ('g3_single.npz', 'g3_single_n10.npz') + tuple(f'g4_ctl_n{n}.npz' for n in (6, 7, 8, 9, 10)):
    g = np.load(os.path.join(golden_dir, fname))
    n = int(g['nbits'])
    with device.DeviceState(n, 128, fusion=fusion) as st:
      for name, gate, out in zip(g['names'], g['gates'], g['outs']):
        parts = str(name).split(':')
        st.upload(g['psi0'])
        if len(parts) == 2:
          st.apply1(gate, int(parts[1]))
        else:
          st.applyc(gate, int(parts[1]), int(parts[2]))
        err = np.max(np.abs(st.download() - out))
        assert err <= TOL, (fname, name, err)


@pytest.mark.parametrize('fusion', FUSIONS)
def test_golden_every_ordered_pair_n9_n10(golden_dir, fusion):
  """G4 of SURVEY 8(c) for every ordered (ctl, tgt) at 9 and 10 qubits against what the reference's applyc computed
  (tests/golden/g4_pairs_n9_n10.npz: 16 probe inner products per case, a few outputs in full)."""
  g = np.load(os.path.join(golden_dir, 'g4_pairs_n9_n10.npz'))
  for n in (9, 10):
    probes = g[f'probes_n{n}']
    full = {str(k): v for k, v in zip(g[f'full_names_n{n}'], g[f'full_n{n}'])}
    with device.DeviceState(n, 128, fusion=fusion) as st:
      for name, gate, proj in zip(g[f'names_n{n}'], g[f'gates_n{n}'], g[f'proj_n{n}']):
        _, c, t = str(name).split(':')
        st.upload(g[f'psi0_n{n}'])
        st.applyc(gate, int(c), int(t))
        got = st.download()
        assert np.max(np.abs(probes.conj() @ got - proj)) <= 1e-13, (n, name)
        if str(name) in full:
          assert np.max(np.abs(got - full[str(name)])) <= TOL, (n, name)


@pytest.mark.parametrize('fusion', FUSIONS)
def test_golden_complex64(golden_dir, fusion):
  g = np.load(os.path.join(golden_dir, 'g7_c64.npz'))
  n = int(g['nbits'])
  with device.DeviceState(n, 64, fusion=fusion) as st:
    for name, gate, out in zip(g['names'], g['gates'], g['outs']):
      parts = str(name).split(':')
      st.upload(g['psi0'])
      if len(parts) == 2:
        st.apply1(gate, int(parts[1]))
      else:
        st.applyc(gate, int(parts[1]), int(parts[2]))
      got = st.download()
      assert got.dtype == np.complex64
      assert np.max(np.abs(got - out)) <= 2e-6, name


@pytest.mark.parametrize('fusion', FUSIONS)
def test_golden_traces(golden_dir, fusion):
  """Recorded native call streams of reference circuits (QFT, supremacy, Grover,
  multi_control, the negative-control sequence of circuit_test.py:97-102)."""
  files = sorted(glob.glob(os.path.join(golden_dir, 'g5_*.npz'))) + [
      os.path.join(golden_dir, 'g1_qft12.npz'), os.path.join(golden_dir, 'py_fallback.npz')]
  assert len(files) >= 10
  for f in files:
    g = np.load(f)
    n = int(g['nbits'])
    with device.DeviceState(n, 128, fusion=fusion) as st:
      st.upload(g['init'])
      st.run_stream(g['ops'], g['gates'])
      got = st.download()
    err = np.max(np.abs(got - g['final']))
    assert err <= 1e-10, (os.path.basename(f), err)
    assert err <= 2e-13, (os.path.basename(f), err)


@pytest.mark.parametrize('fusion', FUSIONS)
@pytest.mark.parametrize('n', [1, 2, 3, 5, 6, 7, 10, 11, 12, 13, 16])
def test_every_target_and_control_vs_oracle(oracle, fusion, n):
  rng = np.random.default_rng(100 + n)
  psi0 = _rand_state(rng, n)
  pool = _gate_pool(rng)
  with device.DeviceState(n, 128, fusion=fusion) as st:
    for t in range(n):
      for gi in rng.choice(len(pool), size=3, replace=False):
        want = psi0.copy()
        oracle.apply1(want, pool[gi], n, t)
        st.upload(psi0)
        st.apply1(pool[gi], t)
        assert np.max(np.abs(st.download() - want)) <= TOL, (n, t, gi)
    pairs = [(c, t) for c in range(n) for t in range(n) if c != t]
    if len(pairs) > 40:
      pairs = [pairs[i] for i in rng.choice(len(pairs), size=40, replace=False)]
    for c, t in pairs:
      gi = int(rng.integers(len(pool)))
      want = psi0.copy()
      oracle.applyc(want, pool[gi], n, c, t)
      st.upload(psi0)
      st.applyc(pool[gi], c, t)
      assert np.max(np.abs(st.download() - want)) <= TOL, (n, c, t, gi)


@pytest.mark.parametrize('fusion', FUSIONS)
@pytest.mark.parametrize('n,ngates,seed', [(4, 60, 0), (9, 150, 1), (12, 300, 2), (15, 400, 3), (20, 250, 4)])
def test_random_streams_vs_oracle(oracle, fusion, n, ngates, seed):
  rng = np.random.default_rng(seed)
  pool = _gate_pool(rng)
  ops, gs = [], []
  for _ in range(ngates):
    t = int(rng.integers(n))
    g = pool[int(rng.integers(len(pool)))]
    if n > 1 and rng.random() < 0.55:
      c = int((t + 1 + rng.integers(n - 1)) % n)
      ops.append((c, t))
    else:
      ops.append((NO_CTL, t))
    gs.append(np.asarray(g, dtype=np.complex128).reshape(4))
  ops = np.array(ops, dtype=np.int32)
  g8 = np.array(gs).view(np.float64).reshape(-1, 8)
  psi0 = _rand_state(rng, n)
  want = psi0.copy()
  oracle.run_stream(want, n, ops, g8)
  with device.DeviceState(n, 128, fusion=fusion) as st:
    st.upload(psi0)
    st.run_stream(ops, g8)
    got = st.download()
    n2 = st.norm2()
  assert np.max(np.abs(got - want)) <= 1e-11
  assert abs(n2 - 1.0) < 1e-11


def test_multi_control_masks_vs_oracle(oracle):
  """qh_apply_bits with several control bits == nested reference semantics."""
  n = 10
  rng = np.random.default_rng(9)
  psi0 = _rand_state(rng, n)
  u = _rand_unitary(rng)
  for fusion in FUSIONS:
    with device.DeviceState(n, 128, fusion=fusion) as st:
      for mask, tbit in ((0b0000000110, 0), (0b1000000001, 5), (0b0101010000, 9), (0b0000111000, 2)):
        want = psi0.copy()
        idx = np.arange(1 << n)
        sel = ((idx & mask) == mask) & (((idx >> tbit) & 1) == 0)
        a, b = want[idx[sel]], want[idx[sel] | (1 << tbit)]
        want[idx[sel]] = u[0, 0] * a + u[0, 1] * b
        want[idx[sel] | (1 << tbit)] = u[1, 0] * a + u[1, 1] * b
        st.upload(psi0)
        st.apply_bits(mask, tbit, u)
        assert np.max(np.abs(st.download() - want)) <= TOL


def test_host_dropin_entry_points(oracle):
  lib = native.load()
  dp = ctypes.POINTER(ctypes.c_double)
  rng = np.random.default_rng(1)
  for bw, dtype, tol in ((128, np.complex128, TOL), (64, np.complex64, 2e-6)):
    n = 9
    psi = _rand_state(rng, n, dtype)
    want = psi.copy()
    g = _rand_unitary(rng)
    g8 = gates.as8(g)
    native.check(lib.qh_host_apply1(psi.ctypes.data, g8.ctypes.data_as(dp), n, 3, bw))
    oracle.apply1(want, g, n, 3)
    native.check(lib.qh_host_applyc(psi.ctypes.data, g8.ctypes.data_as(dp), n, 8, 0, bw))
    oracle.applyc(want, g, n, 8, 0)
    native.check(lib.qh_host_applyc(psi.ctypes.data, g8.ctypes.data_as(dp), n, -2, 5, bw))
    oracle.applyc(want, g, n, -2, 5)
    assert np.max(np.abs(psi - want)) <= tol


def test_host_dropin_scratch_lifetime(oracle):
  """qh_host_* keep per-thread device scratch for the two most recent register shapes; alternating
  register sizes, a third shape, qh_host_release() and a second thread all give the oracle's result."""
  import threading
  lib = native.load()
  dp = ctypes.POINTER(ctypes.c_double)
  rng = np.random.default_rng(2)
  g = _rand_unitary(rng)
  g8 = gates.as8(g)

  def one(n, bw=128):
    psi = _rand_state(rng, n, np.complex128 if bw == 128 else np.complex64)
    want = psi.copy()
    native.check(lib.qh_host_apply1(psi.ctypes.data, g8.ctypes.data_as(dp), n, n - 1, bw))
    oracle.apply1(want, g, n, n - 1)
    native.check(lib.qh_host_applyc(psi.ctypes.data, g8.ctypes.data_as(dp), n, 0, n - 2, bw))
    oracle.applyc(want, g, n, 0, n - 2)
    assert np.max(np.abs(psi - want)) <= (TOL if bw == 128 else 2e-6)
  for n in (10, 12, 10, 12, 14, 10, 12):
    one(n)
  one(10, 64)
  assert lib.qh_host_release() == 0
  assert lib.qh_host_release() == 0          # idempotent
  one(12)
  errs = []

  def worker():
    try:
      one(11)
      lib.qh_host_release()
    except Exception as e:  # pylint: disable=broad-except
      errs.append(e)
  t = threading.Thread(target=worker)
  t.start()
  t.join()
  assert not errs
  lib.qh_host_release()


def test_readers(oracle):
  n = 14
  rng = np.random.default_rng(4)
  psi = _rand_state(rng, n)
  with device.DeviceState(n, 128) as st:
    st.upload(psi)
    assert abs(st.norm2() - 1.0) < 1e-12
    idx, p = st.argmax()
    assert idx == int(np.argmax(np.abs(psi))) and abs(p - np.abs(psi[idx]) ** 2) < 1e-15
    for bit in (0, 5, 13):
      want = float(np.sum(np.abs(psi[((np.arange(1 << n) >> bit) & 1) == 1]) ** 2))
      assert abs(st.prob_bit(bit) - want) < 1e-12
    st.project_bit(5, 1)
    proj = psi.copy()
    proj[((np.arange(1 << n) >> 5) & 1) == 0] = 0
    assert np.max(np.abs(st.download() - proj)) == 0
    st.scale(2.0 - 1.0j)
    assert np.max(np.abs(st.download() - proj * (2.0 - 1.0j))) < 1e-15
    st.init_basis(77)
    got = st.download()
    assert got[77] == 1 and np.count_nonzero(got) == 1


def _random_stream(rng, n, ngates):
  ops, gs = [], []
  for _ in range(ngates):
    q = np.linalg.qr(rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)))[0]
    if rng.random() < 0.35:
      c, t = (int(v) for v in rng.choice(n, 2, replace=False))
      g = np.diag([1.0, np.exp(1j * rng.uniform(0, 6.28))]) if rng.random() < 0.5 else q
      ops.append((c, t))
    else:
      g = q
      ops.append((workloads.NO_CTL, int(rng.integers(0, n))))
    gs.append(np.asarray(g, dtype=np.complex128).reshape(4))
  return np.array(ops, dtype=np.int32), np.array(gs).view(np.float64).reshape(-1, 8)


@pytest.mark.parametrize('relayout', ['1', '0'])
def test_argmax_behind_a_flush_uses_the_last_sweep(oracle, monkeypatch, relayout):
  """qh_argmax right behind a fused flush takes the per-unit maxima the last sweep left (SweepParams::tilemax, engine.hip
  argmax_from_tilemax) instead of a full pass: the same index and probability as the full pass (QH_FUSED_ARGMAX=0) and as
  NumPy on the downloaded state, for random circuits in whatever layout relayout sweeps leave (and in place), for exact ties
  -- the smallest LOGICAL index wins, state.py:60-78 -- and for a uniform state (every unit ties: the full pass decides)."""
  monkeypatch.setenv('QH_RELAYOUT', relayout)
  rng = np.random.default_rng(31)
  for case in range(14):
    n = int(rng.integers(10, 21))
    bw = 128 if case < 10 else 64             # complex64 tiles too: their probabilities are doubles, as in the full pass
    ops, g8 = _random_stream(rng, n, int(rng.integers(5, 120)))
    res = {}
    for fused in ('1', '0'):
      monkeypatch.setenv('QH_FUSED_ARGMAX', fused)
      with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
        st.init_basis(int(rng.integers(0, 1 << n)) if fused == '1' else res['init'])
        if fused == '1':
          res['init'] = int(np.argmax(np.abs(st.download())))
        st.run_stream(ops, g8)
        idx, p = st.argmax()
        psi = st.download()
        res[fused] = (idx, p)
      pr = psi.real.astype(np.float64) ** 2 + psi.imag.astype(np.float64) ** 2
      want = int(np.argmax(pr))
      assert abs(p - pr[want]) <= 4e-16 * pr[want] + 1e-300 and pr[idx] >= pr[want] * (1 - 4e-16), (case, fused, idx, want)
    assert res['1'] == res['0'], (case, res)
  monkeypatch.setenv('QH_FUSED_ARGMAX', '1')
  # exact ties: (|a> + |b>)/sqrt(2) pushed through a circuit of permutations / diagonal gates only -- the two peaks stay exactly
  # equal, relayout sweeps reorder the physical indices, the smaller LOGICAL index must win
  n = 16
  xg = np.array([0, 1, 1, 0], dtype=np.complex128)
  for trial in range(6):
    a, b = sorted(int(v) for v in rng.choice(1 << n, 2, replace=False))
    psi0 = np.zeros(1 << n, dtype=np.complex128)
    psi0[a] = psi0[b] = 1 / np.sqrt(2)
    ops, gs = [], []
    for _ in range(60):
      t = int(rng.integers(0, n))
      if rng.random() < 0.5:
        ops.append((workloads.NO_CTL, t)); gs.append(xg)
      else:
        c = int((t + 1 + rng.integers(0, n - 1)) % n)
        ops.append((c, t)); gs.append(np.array([1, 0, 0, (1, 1j, -1, -1j)[int(rng.integers(0, 4))]], dtype=np.complex128))   # (exact: the peaks stay equal)
    ops = np.array(ops, dtype=np.int32)
    g8 = np.array(gs).view(np.float64).reshape(-1, 8)
    want = psi0.copy()
    oracle.run_stream(want, n, ops, g8)
    with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
      st.upload(psi0)
      st.run_stream(ops, g8)
      idx, p = st.argmax()
    peaks = np.flatnonzero(np.abs(want) > 0.5)
    assert len(peaks) == 2 and idx == int(peaks[0]) and abs(p - 0.5) < 1e-15, (trial, idx, peaks)
  # uniform state: every amplitude ties, index 0 wins
  hg = np.array([1, 1, 1, -1], dtype=np.complex128) / np.sqrt(2)
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(0)
    st.run_stream(np.array([(workloads.NO_CTL, q) for q in range(n)], dtype=np.int32), np.tile(hg.view(np.float64), (n, 1)))
    idx, p = st.argmax()
    psi = st.download()
  assert idx == int(np.argmax(np.abs(psi) ** 2)) and abs(p - 2.0 ** -n) < 1e-18


def test_supremacy20_sampled_golden(golden_dir):
  """G9: the reference's 20-qubit depth-20 supremacy run (trace + sampled amplitudes)."""
  g = np.load(os.path.join(golden_dir, 'g9_supremacy_n20_s0.npz'))
  n = int(g['nbits'])
  for fusion in FUSIONS:
    with device.DeviceState(n, 128, fusion=fusion) as st:
      st.init_basis(int(g['init_index']))
      st.run_stream(g['ops'], g['gates'])
      got = st.download()
    assert np.max(np.abs(got[g['idx']] - g['amp'])) <= 1e-12
    assert abs(np.vdot(got, got).real - float(g['norm2'])) < 1e-11


@pytest.mark.parametrize('nq', [22, 24, 26])
def test_qft22_sampled_golden(golden_dir, nq):
  g = np.load(os.path.join(golden_dir, f'g6_qft{nq}.npz'))
  n, x = int(g['nbits']), int(g['x'])
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  for fusion in FUSIONS:
    with device.DeviceState(n, 128, fusion=fusion) as st:
      st.init_basis(x)
      st.run_stream(ops, g8)
      got = st.download()
    assert np.max(np.abs(got[g['idx']] - g['amp'])) <= 1e-12
    assert abs(np.vdot(got, got).real - 1) < 1e-12


@pytest.mark.parametrize('fusion', FUSIONS)
def test_full_size_30q_properties(fusion):
  """BASELINE config 2 size: 30-qubit QFT.  The reference cannot be run at this
  size in seconds, so: closed form on sampled indices, norm, and QFT followed by
  the inverse circuit returns the basis state (size-independent properties)."""
  n = 30
  x = 0x2CB9A5E3 & ((1 << n) - 1)
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  with device.DeviceState(n, 128, fusion=fusion) as st:
    st.init_basis(x)
    st.run_stream(ops, g8)
    assert abs(st.norm2() - 1.0) < 1e-10
    rng = np.random.default_rng(30)
    idx = rng.integers(0, 1 << n, size=512)
    want = workloads.qft_analytic(n, x, idx)
    got = np.array([st.amplitude(int(i)) for i in idx])
    assert np.max(np.abs(got - want)) <= 1e-10
    # a contiguous window too (exercises low bits)
    win = st.download(offset=(1 << 29) + 12345, count=4096)
    wantw = workloads.qft_analytic(n, x, np.arange((1 << 29) + 12345, (1 << 29) + 12345 + 4096))
    assert np.max(np.abs(win - wantw)) <= 1e-10
    # inverse: adjoint gates in reverse order
    inv_ops = ops[::-1].copy()
    inv_g = g8[::-1].copy().reshape(-1, 4, 2)
    inv_c = inv_g[..., 0] + 1j * inv_g[..., 1]
    inv_c = np.conj(inv_c.reshape(-1, 2, 2).transpose(0, 2, 1)).reshape(-1, 4)
    st.run_stream(inv_ops, np.ascontiguousarray(inv_c).view(np.float64).reshape(-1, 8))
    i, p = st.argmax()
    assert i == x and abs(p - 1.0) < 1e-10
    assert abs(st.norm2() - 1.0) < 1e-10


def test_error_behaviour_on_device():
  with device.DeviceState(6, 128) as st:
    with pytest.raises(native.QhError) as e:
      st.apply1(gates.hadamard(), 6)
    assert e.value.code == native.QH_ERR_BAD_QUBIT
    with pytest.raises(native.QhError) as e:
      st.applyc(gates.hadamard(), 2, 2)
    assert e.value.code == native.QH_ERR_SAME_QUBIT
    st.init_basis(0)
    st.apply1(gates.hadamard(), 0)  # still usable after errors
    assert abs(st.norm2() - 1) < 1e-15


def test_many_common_controls_fused(oracle):
  """5-control gates: more common control bits than the tile enumeration can fix."""
  n = 14
  rng = np.random.default_rng(21)
  psi0 = _rand_state(rng, n)
  u = _rand_unitary(rng)
  mask = (1 << 13) | (1 << 12) | (1 << 11) | (1 << 9) | (1 << 8) | (1 << 7)
  want = psi0.copy()
  idx = np.arange(1 << n)
  for tbit in (10, 2):
    sel = ((idx & mask) == mask) & (((idx >> tbit) & 1) == 0)
    a, b = want[idx[sel]].copy(), want[idx[sel] | (1 << tbit)].copy()
    want[idx[sel]] = u[0, 0] * a + u[0, 1] * b
    want[idx[sel] | (1 << tbit)] = u[1, 0] * a + u[1, 1] * b
  ph = np.exp(0.3j)
  want[(idx & (mask | 1 << 10)) == (mask | 1 << 10)] *= ph
  for fusion in FUSIONS:
    with device.DeviceState(n, 128, fusion=fusion) as st:
      st.upload(psi0)
      st.apply_bits(mask, 10, u)
      st.apply_bits(mask, 2, u)
      st.apply_bits(mask, 10, gates.u1(0.3))
      assert np.max(np.abs(st.download() - want)) <= TOL


@pytest.mark.parametrize('n,ngates,seed', [(8, 120, 0), (11, 250, 1), (14, 300, 2), (17, 200, 3)])
def test_complex64_fused_streams_vs_oracle(oracle, n, ngates, seed):
  """complex64 states through the fused sweeps (sweep_island_f32_*): reference default width."""
  rng = np.random.default_rng(seed + 50)
  pool = _gate_pool(rng)
  ops, gs = [], []
  for _ in range(ngates):
    t = int(rng.integers(n))
    g = pool[int(rng.integers(len(pool)))]
    if rng.random() < 0.55:
      ops.append((int((t + 1 + rng.integers(n - 1)) % n), t))
    else:
      ops.append((NO_CTL, t))
    gs.append(np.asarray(g, dtype=np.complex128).reshape(4))
  ops = np.array(ops, dtype=np.int32)
  g8 = np.array(gs).view(np.float64).reshape(-1, 8)
  psi0 = _rand_state(rng, n, np.complex64)
  want = psi0.copy()
  for (c, t), g in zip(ops, gs):
    if c == NO_CTL:
      oracle.apply1(want, g.astype(np.complex64), n, int(t))
    else:
      oracle.applyc(want, g.astype(np.complex64), n, int(c), int(t))
  with device.DeviceState(n, 64, fusion=native.QH_FUSE_SWEEP) as st:
    st.upload(psi0)
    st.run_stream(ops, g8)
    got = st.download()
    s = st.stats()
  assert s['sweeps'] >= 1                      # really the fused path
  assert got.dtype == np.complex64
  assert np.max(np.abs(got - want)) <= 3e-5    # float32 accumulation over hundreds of gates


@pytest.mark.parametrize('n,bw', [(20, 64), (27, 64), (27, 128)])
def test_qft_fused_analytic(n, bw):
  """QFT of a basis state against the closed form, every amplitude: 20 qubits in one or two
  sweeps; 27 qubits with super-tiles (wave bits, OP_WSWAP) in both element widths."""
  x = 0xBEEF5 & ((1 << n) - 1)
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(x)
    st.run_stream(ops, g8)
    got = st.download()
    sweeps = st.stats()['sweeps']
  assert sweeps <= 4
  tol = (3e-5 if bw == 64 else 1e-12) * 2.0 ** (-n / 2)      # relative to the amplitude modulus 2^(-n/2)
  for lo in range(0, 1 << n, 1 << 22):                     # closed form in slices (python-int phases)
    idx = np.arange(lo, min(lo + (1 << 22), 1 << n), 4099 if n > 22 else 1)
    want = workloads.qft_analytic(n, x, idx)
    assert np.max(np.abs(got[idx] - want)) < tol


@pytest.mark.parametrize('lane_valu', ['1', '2'])
@pytest.mark.parametrize('bw', [128, 64])
def test_unit_entry_butterflies_vs_oracle(oracle, bw, lane_valu, monkeypatch):
  """h / yroot / v and adjoints run as add-only butterflies inside a sweep (planner.h
  settle_butterflies, island L_bf*); their scalars land in one other gate.  Every
  variant on every bit of a 13-qubit state (lane bits, split-lane bits, register
  bits), next to a general gate, and all-butterfly sweeps (the sink is a butterfly).
  lane_valu=2 forces the no-LDS lane paths (DPP partner fetch on lane bits 0..3,
  v_permlane swaps for lane bits 4/5) that the planner otherwise picks by cost model."""
  monkeypatch.setenv('QH_LANE_VALU', lane_valu)
  n = 13
  dt = np.complex128 if bw == 128 else np.complex64
  rng = np.random.default_rng(99)
  had, yr, v = gates.hadamard(), gates.yroot(), gates.vgate()
  variants = [had, yr, np.conj(yr.reshape(2, 2).T), v, np.conj(np.asarray(v).reshape(2, 2).T), -0.3j * np.asarray(had)]
  streams = []
  for g in variants:
    for t in range(n):
      streams.append([(NO_CTL, t, g), (NO_CTL, (t + 5) % n, gates.rx(0.7)), (NO_CTL, (t + 1) % n, g)])
  streams.append([(NO_CTL, t, variants[t % 5]) for t in range(n)])          # butterflies only
  streams.append([(NO_CTL, t, had) for t in range(n)] + [((t + 1) % n, t, gates.u1(0.2 * t)) for t in range(n)] +
                 [(NO_CTL, t, v) for t in range(n)])
  worst = 0.0
  with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
    for stream in streams:
      psi0 = _rand_state(rng, n, dt)
      want = psi0.copy()
      for c, t, g in stream:
        g = np.asarray(g, dtype=dt).reshape(4)
        if c == NO_CTL:
          oracle.apply1(want, g, n, t)
        else:
          oracle.applyc(want, g, n, c, t)
      st.upload(psi0)
      for c, t, g in stream:
        if c == NO_CTL:
          st.apply1(g, t)
        else:
          st.applyc(g, c, t)
      worst = max(worst, float(np.max(np.abs(st.download() - want))))
  assert worst <= (TOL if bw == 128 else 3e-6), worst


@pytest.mark.parametrize('bw', [128, 64])
def test_init_product_matches_kron(bw):
  """qh_init_product == np.kron of the factors (circuit.py:121-164)."""
  rng = np.random.default_rng(5)
  dt = np.complex128 if bw == 128 else np.complex64

  def rnd(n):
    return _rand_state(rng, n)

  factors = [(3, 5), (2, rnd(2)), (10, rnd(10)), (1, np.array([0.6, 0.8j])), (4, 9)]
  want = np.ones(1, dtype=np.complex128)
  for n, x in factors:
    if isinstance(x, int):
      t = np.zeros(1 << n, dtype=np.complex128)
      t[x] = 1
    else:
      t = x
    want = np.kron(want, t)
  with device.DeviceState(20, bw, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_product(factors)
    got = st.download()
    assert got.dtype == dt
    assert np.max(np.abs(got - want.astype(dt))) <= (1e-15 if bw == 128 else 1e-7)
    assert abs(st.norm2() - 1.0) < 1e-6
    # 33 single-qubit factors are merged on the host before they reach the C-ABI ...
    many = [(1, np.array([0.6, 0.8]))] * 20
    st.init_product(many)
    assert abs(st.amplitude(0) - 0.6 ** 20) < (1e-12 if bw == 128 else 1e-7)
    # ... and a factor list that does not add up is rejected
    with pytest.raises(native.QhError):
      st.init_product([(3, 1), (5, 2)])


@pytest.mark.parametrize('wave_bits', ['0', '2'])
@pytest.mark.parametrize('n,ngates,seed', [(15, 400, 3), (18, 300, 7)])
def test_super_tile_variants_vs_oracle(oracle, monkeypatch, wave_bits, n, ngates, seed):
  """The default plan uses one wave bit per tile (workgroups of two waves exchanging through
  LDS, OP_WSWAP); two wave bits (four waves) and none are the other supported shapes."""
  monkeypatch.setenv('QH_WAVE_BITS', wave_bits)
  rng = np.random.default_rng(seed)
  pool = _gate_pool(rng)
  ops, gs = [], []
  for _ in range(ngates):
    t = int(rng.integers(n))
    g = pool[int(rng.integers(len(pool)))]
    if rng.random() < 0.5:
      ops.append((int((t + 1 + rng.integers(n - 1)) % n), t))
    else:
      ops.append((NO_CTL, t))
    gs.append(np.asarray(g, dtype=np.complex128).reshape(4))
  ops = np.array(ops, dtype=np.int32)
  g8 = np.array(gs).view(np.float64).reshape(-1, 8)
  psi0 = _rand_state(rng, n)
  want = psi0.copy()
  oracle.run_stream(want, n, ops, g8)
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.upload(psi0)
    st.run_stream(ops, g8)
    got = st.download()
  assert np.max(np.abs(got - want)) <= 1e-11


@pytest.mark.parametrize('bw', [64, 128])
def test_long_butterfly_runs_stay_in_range(oracle, bw):
  """Hundreds of h / v / yroot in ONE sweep: their scalars are moved into sink gates, so the
  stored amplitudes drift by sqrt(2) per gate in between -- intermediate sinks keep them in
  the range of the element type (found by tools/fuzz_parity.py: complex64 overflowed)."""
  n, ngates = 11, 600
  dt = np.complex128 if bw == 128 else np.complex64
  rng = np.random.default_rng(2024)
  pool = [gates.hadamard(), gates.vgate(), gates.yroot()]
  psi0 = _rand_state(rng, n, dt)
  want = psi0.copy()
  with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
    st.upload(psi0)
    for _ in range(ngates):
      g = np.asarray(pool[int(rng.integers(3))], dtype=np.complex128).reshape(4)
      t = int(rng.integers(n))
      oracle.apply1(want, g.astype(dt), n, t)
      st.apply1(g, t)
    got = st.download()
    assert st.stats()['sweeps'] == 1
  assert np.all(np.isfinite(got))
  assert np.max(np.abs(got - want)) <= (1e-11 if bw == 128 else 2e-4)


@pytest.mark.parametrize('bw', [128, 64])
@pytest.mark.parametrize('lane_valu', ['1', '2'])
def test_qft_of_random_states_small_tiles_vs_oracle(oracle, monkeypatch, bw, lane_valu):
  """QFT and inverse QFT on random states of 8..17 qubits against the oracle: the phase ladders run as factor
  trees (DG_BITFAC) on tiles with 2, 3, 4 and 5 (complex64: 6) register bits, whose handlers are separate code,
  with the lane butterflies on either path."""
  monkeypatch.setenv('QH_LANE_VALU', lane_valu)
  rng = np.random.default_rng(77)
  for n in range(8, 18):
    ops, g8 = workloads.qft_stream(range(n)).arrays()
    dt = np.complex128 if bw == 128 else np.complex64
    psi0 = _rand_state(rng, n, dt)
    want = psi0.astype(np.complex128)
    oracle.run_stream(want, n, ops, g8)
    with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
      st.upload(psi0)
      st.run_stream(ops, g8)
      got = st.download()
      inv_ops = ops[::-1].copy()
      z = g8[::-1].copy().view(np.complex128).reshape(-1, 2, 2)
      gi = np.ascontiguousarray(np.conj(z.transpose(0, 2, 1))).view(np.float64).reshape(-1, 8)
      st.run_stream(inv_ops, gi)
      back = st.download()
    tol = 1e-12 if bw == 128 else 2e-5
    assert np.max(np.abs(got - want)) <= tol, (n, bw)
    assert np.max(np.abs(back - psi0)) <= 2 * tol, (n, bw)


@pytest.mark.parametrize('bw', [128, 64])
def test_per_gate_kernel_shapes_every_target_at_22_qubits(oracle, bw):
  """Round 4: the per-gate kernels pick their launch shape by target bit (kernels_gate.hip.h: k_pair_line for targets
  inside the 128-byte line -- partner by DPP moves --, k_pair_tile / k_diag_tile wave tiles, U = 16 tiles for bits
  20..25; engine.hip launch_pair / launch_diag).  Every target of a 22-qubit state, uncontrolled and under controls
  on low, middle and high bits (inserted control bits below and above the target, the lowpred predicate on bits 0-1),
  dense / real / anti-diagonal / diagonal gates, both widths -- unfused, against the oracle."""
  n = 22
  rng = np.random.default_rng(2200 + bw)
  dtype = np.complex128 if bw == 128 else np.complex64
  tol = TOL if bw == 128 else 2e-6
  psi0 = _rand_state(rng, n, dtype)
  dense = [_rand_unitary(rng), gates.hadamard(), gates.pauli_x(), gates.pauli_y(), gates.ry(0.7)]
  diag = [gates.u1(0.37), gates.pauli_z(), gates.rz(0.9), gates.tgate()]
  with device.DeviceState(n, bw, fusion=native.QH_FUSE_OFF) as st:
    def check(apply_dev, apply_ref, what):
      want = psi0.astype(np.complex128)
      apply_ref(want)
      st.upload(psi0)
      apply_dev()
      err = float(np.max(np.abs(st.download().astype(np.complex128) - want)))
      assert err <= tol, (bw, what, err)
    for q in range(n):                       # reference qubit q = index bit n-1-q
      g = dense[q % len(dense)]
      check(lambda: st.apply1(g, q), lambda w: oracle.apply1(w, g, n, q), ('apply1', q))
      d = diag[q % len(diag)]
      check(lambda: st.apply1(d, q), lambda w: oracle.apply1(w, d, n, q), ('apply1-diag', q))
      for c in {(q + 1) % n, (q + 11) % n, n - 1, n - 2, n - 3, 0} - {q}:
        check(lambda: st.applyc(g, c, q), lambda w: oracle.applyc(w, g, n, c, q), ('applyc', c, q))
        check(lambda: st.applyc(d, c, q), lambda w: oracle.applyc(w, d, n, c, q), ('applyc-diag', c, q))
    # several controls at once through qh_apply_bits (controls on index bits 0, 1 -> lane predicate; 2.. -> inserted)
    idx = np.arange(1 << n, dtype=np.uint64)
    for tgt_bit, cbits in ((0, (1, 2, 9)), (1, (0, 21)), (2, (0, 1, 3)), (5, (0, 2, 20)), (21, (0, 1, 2)), (20, (3, 21)), (12, (1, 13, 19))):
      g = dense[tgt_bit % len(dense)]
      cm = sum(1 << b for b in cbits)
      def ref(w, g=g, cm=cm, tgt_bit=tgt_bit):
        tmp = w.copy()
        oracle.apply1(tmp, g, n, n - 1 - tgt_bit)
        sel = (idx & np.uint64(cm)) == np.uint64(cm)
        w[sel] = tmp[sel]
      check(lambda: st.apply_bits(cm, tgt_bit, g), ref, ('apply_bits', cbits, tgt_bit))


@pytest.mark.parametrize('n', [13, 15])
@pytest.mark.parametrize('bw', [128, 64])
def test_fused_phases_and_inlined_groups_vs_oracle(oracle, bw, n):
  """Round 4: (a) a T / T^+ / odd multiple of pi/4 waiting on a butterfly's target rides in the butterfly (planner.h
  OPF_ROT_*, island L_bfr*: every variant x every tile position of the target, both rotations, both signs of the scale),
  next to phases that must NOT fuse (s, z, a general u1, a phase under a control); (b) DIAG ops of one simple group are
  folded into their op headers (island L_d1*): sign flips (cz), uniform and lane-masked factors on every one- and two-bit
  register mask the plan produces, and masks of three bits (ccz-like).  complex64 takes the general paths: same results."""
  dt = np.complex128 if bw == 128 else np.complex64
  rng = np.random.default_rng(400 + n)
  had, yr, v, t8 = gates.hadamard(), gates.yroot(), gates.vgate(), gates.tgate()
  dag = lambda g: np.conj(np.asarray(g).reshape(2, 2).T)
  z = np.diag([1.0, -1.0]).astype(np.complex128)
  bflies = [had, yr, dag(yr), v, dag(v)]
  phases = [t8, dag(t8), np.asarray(t8).reshape(2, 2) @ np.asarray(t8).reshape(2, 2) @ np.asarray(t8).reshape(2, 2),   # T, T^+, T^3
            gates.sgate(), z, gates.u1(0.3), -1.0 * np.asarray(t8).reshape(2, 2)]                                # not fused: S, Z, U1; scaled T
  streams = []
  for t in range(n):                       # every position of the target: lanes, registers, wave bits
    s = [(NO_CTL, q, had) for q in range(n)]
    for k, b in enumerate(bflies):
      s += [(NO_CTL, t, phases[(t + k) % len(phases)]), (NO_CTL, t, b)]
    s += [((t + 1) % n, t, t8), (NO_CTL, t, v)]                  # a controlled T is a two-bit phase: stays a group
    streams.append(s)
  for a in range(0, n, 2):                 # single groups: cz on every pair class, phases on single bits, three-bit masks
    s = [(NO_CTL, q, [had, v, yr][q % 3]) for q in range(n)]
    for b in range(n):
      if b != a:
        s += [(a, b, z), (NO_CTL, (a + b) % n, [had, v][b % 2])]
    s += [(NO_CTL, a, gates.rz(0.4 + a)), (NO_CTL, (a + 1) % n, had), (NO_CTL, a, gates.u1(0.2)), (NO_CTL, (a + 2) % n, yr)]
    streams.append(s)
  worst = 0.0
  with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
    for stream in streams:
      psi0 = _rand_state(rng, n, dt)
      want = psi0.copy()
      for c, t, g in stream:
        g = np.asarray(g, dtype=dt).reshape(4)
        if c == NO_CTL:
          oracle.apply1(want, g, n, t)
        else:
          oracle.applyc(want, g, n, c, t)
      st.upload(psi0)
      for c, t, g in stream:
        if c == NO_CTL:
          st.apply1(g, t)
        else:
          st.applyc(g, c, t)
      worst = max(worst, float(np.max(np.abs(st.download() - want))))
    # three-bit register masks (the "any other mask" handlers): doubly-controlled z / phase on register-bit qubits
    masks = native.load()
    psi0 = _rand_state(rng, n, dt)
    want = psi0.copy()
    st.upload(psi0)
    idx = np.arange(1 << n, dtype=np.uint64)
    for (qa, qb, qc), ph in (((n - 8, n - 9, n - 10), -1.0), ((n - 7, n - 9, n - 11), np.exp(0.7j)), ((n - 8, n - 10, n - 11), -1.0)):
      sel = np.ones(1 << n, dtype=bool)
      cm = 0
      for q_ in (qa, qb):
        sel &= ((idx >> np.uint64(n - 1 - q_)) & np.uint64(1)).astype(bool)
        cm |= 1 << (n - 1 - q_)
      sel &= ((idx >> np.uint64(n - 1 - qc)) & np.uint64(1)).astype(bool)
      want[sel] *= dt(ph) if bw == 64 else ph
      g8 = np.array([1, 0, 0, 0, 0, 0, np.real(ph), np.imag(ph)], dtype=np.float64)
      native.check(masks.qh_apply_bits(st.h, cm, n - 1 - qc, g8.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
      st.apply1(had, (qa + 3) % n)
      oracle.apply1(want, np.asarray(had, dtype=dt).reshape(4), n, (qa + 3) % n)
    worst = max(worst, float(np.max(np.abs(st.download() - want))))
  assert worst <= (TOL if bw == 128 else 3e-6), worst
