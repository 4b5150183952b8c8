"""Relayout sweeps on the GPU: a handle that owns its memory gathers scattered tiles into
a second buffer and keeps a logical->physical bit map; everything a caller can observe must be unchanged.
(The parity suite runs with relayout on by default; these tests aim at the seams: the bit map, the entry
points that restore canonical order, state construction under a permuted map, both widths, the off switch.)"""
import ctypes
import os

import numpy as np
import pytest

from qcc_amd import device, gates, native, workloads
from tests.oracle_lib import NO_CTL

pytestmark = pytest.mark.gpu


def _bitmap(st, n):
  bm = (ctypes.c_int32 * n)()
  native.check(st.lib.qh_get_bitmap(st.h, bm))
  return list(bm)


def _high_bit_circuit(n, seed, count=60):
  """dense gates on HIGH index bits (scattered tiles) mixed with controlled phases everywhere"""
  rng = np.random.default_rng(seed)
  pool = [gates.hadamard(), gates.vgate(), gates.yroot(), gates.ry(0.4), gates.rx(0.7)]
  ops, gs = [], []
  for _ in range(count):
    if rng.random() < 0.5:
      t = int(rng.integers(0, n // 2))            # qubit numbers 0..n/2 = high index bits
      ops.append((NO_CTL, t))
      gs.append(np.asarray(pool[int(rng.integers(len(pool)))], dtype=np.complex128).reshape(4))
    else:
      c, t = (int(v) for v in rng.choice(n, size=2, replace=False))
      ops.append((c, t))
      gs.append(np.asarray(gates.u1(float(rng.uniform(0, 3))), dtype=np.complex128).reshape(4))
  return np.array(ops, dtype=np.int32), np.array(gs).view(np.float64).reshape(-1, 8)


@pytest.mark.parametrize('bw', [128, 64])
def test_relayout_is_invisible_to_every_reader(oracle, bw):
  n = 22
  ops, g8 = _high_bit_circuit(n, 7)
  rng = np.random.default_rng(1)
  psi0 = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
  psi0 /= np.linalg.norm(psi0)
  want = psi0.copy()
  oracle.run_stream(want, n, ops, g8)
  dtype = np.complex128 if bw == 128 else np.complex64
  tol = 1e-12 if bw == 128 else 5e-6
  with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
    st.upload(psi0.astype(dtype))
    st.run_stream(ops, g8)
    st.flush()
    bm = _bitmap(st, n)
    assert sorted(bm) == list(range(n))
    assert bm != list(range(n)), 'this circuit has scattered tiles: some sweep should have re-laid the state out'
    assert bm[:3] == [0, 1, 2] or bw == 64            # the 128-byte line bits never move
    # readers that work in the permuted layout
    idx, p = st.argmax()
    k = int(np.argmax(np.abs(want)))
    assert idx == k and abs(p - abs(want[k]) ** 2) < tol
    for i in (0, 1, k, (1 << n) - 1, 0x15A5A5 & ((1 << n) - 1)):
      assert abs(st.amplitude(i) - want[i]) < tol
    sel = ((np.arange(1 << n) >> 17) & 1) == 1
    assert abs(st.prob_bit(17, 1) - float(np.sum(np.abs(want[sel]) ** 2))) < 10 * tol
    assert _bitmap(st, n) == bm                        # none of them moved anything
    # a window in physical order: canonical order is restored first
    got = st.download(12345, 4096)
    assert np.max(np.abs(got - want[12345:12345 + 4096])) < tol
    assert _bitmap(st, n) == list(range(n))
    # more gates, then the whole state
    st.run_stream(ops[:25], g8[:25])
    oracle.run_stream(want, n, ops[:25], g8[:25])
    assert np.max(np.abs(st.download() - want)) < 2 * tol


def test_state_construction_and_projection_under_a_permuted_map(oracle):
  n = 20
  ops, g8 = _high_bit_circuit(n, 11, 40)
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(77)
    st.run_stream(ops, g8)
    st.flush()
    assert _bitmap(st, n) != list(range(n))
    # basis / product states are written through the current map
    st.init_basis(0x9A5E3 & ((1 << n) - 1))
    st.run_stream(ops[:10], g8[:10])
    want = np.zeros(1 << n, dtype=np.complex128)
    want[0x9A5E3 & ((1 << n) - 1)] = 1
    oracle.run_stream(want, n, ops[:10], g8[:10])
    assert np.max(np.abs(st.download() - want)) < 1e-12
    st.run_stream(ops, g8)
    st.flush()
    f = [(1, np.array([0.6, 0.8j])), (n - 3, 0b1011), (2, np.array([0.5, 0.5, -0.5, 0.5j]))]
    st.init_product(f)
    v = np.kron(np.kron(np.array([0.6, 0.8j]), np.eye(1 << (n - 3))[0b1011]), np.array([0.5, 0.5, -0.5, 0.5j]))
    st.run_stream(ops[:12], g8[:12])
    oracle.run_stream(v, n, ops[:12], g8[:12])
    # projection + rescale in the permuted layout
    st.flush()
    p1 = st.prob_bit(n - 1, 1)
    st.project_bit(n - 1, 1)
    st.scale(1 / np.sqrt(p1))
    sel = ((np.arange(1 << n) >> (n - 1)) & 1) == 1
    assert abs(p1 - np.sum(np.abs(v[sel]) ** 2)) < 1e-12
    v[~sel] = 0
    v /= np.sqrt(p1)
    assert np.max(np.abs(st.download() - v)) < 1e-12


def test_relayout_off_gives_the_same_state_in_place(oracle):
  n = 21
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  res = {}
  for flag in ('1', '0'):
    os.environ['QH_RELAYOUT'] = flag
    try:
      with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
        st.init_basis(0x15A5A5 & ((1 << n) - 1))
        for _ in range(3):                  # the layout cycles from flush to flush
          st.run_stream(ops, g8)
          st.flush()
        res[flag] = (st.download(), st.stats()['kernels_launched'])
        if flag == '0':
          assert _bitmap(st, n) == list(range(n))
    finally:
      del os.environ['QH_RELAYOUT']
  want = np.zeros(1 << n, dtype=np.complex128)
  want[0x15A5A5 & ((1 << n) - 1)] = 1
  for _ in range(3):
    oracle.run_stream(want, n, ops, g8)
  assert np.max(np.abs(res['1'][0] - want)) < 1e-11 and np.max(np.abs(res['0'][0] - want)) < 1e-11


def test_device_pointer_is_canonical():
  n = 18
  ops, g8 = _high_bit_circuit(n, 3, 30)
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(5)
    st.run_stream(ops, g8)
    ref = st.download().copy()
    st.run_stream(ops, g8)
    st.flush()
    _ = st.device_ptr                      # restores canonical order, like download
    assert _bitmap(st, n) == list(range(n))
    with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st2:
      st2.upload(ref)
      st2.run_stream(ops, g8)
      assert np.max(np.abs(st2.download() - st.download())) < 1e-12


def test_device_ptr_stays_valid_across_later_gates(oracle):
  """qh_device_ptr hands the raw pointer out (torch / dlpack interop fetches it once): queued gates run first, the
  layout is canonical, and LATER gates must neither move the state to the other buffer nor permute it (ADVICE r2:
  a relayout sweep after the call used to leave the caller with the stale scratch buffer)."""
  n = 22
  ops, g8 = _high_bit_circuit(n, 11)
  more_ops, more_g = _high_bit_circuit(n, 12)
  rng = np.random.default_rng(2)
  psi0 = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
  psi0 /= np.linalg.norm(psi0)
  want1 = psi0.copy()
  oracle.run_stream(want1, n, ops, g8)
  want2 = want1.copy()
  oracle.run_stream(want2, n, more_ops, more_g)
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.upload(psi0)
    st.run_stream(ops[:30], g8[:30])
    st.flush()                                   # relayout sweeps have run: the layout is permuted now
    assert _bitmap(st, n) != list(range(n))
    st.run_stream(ops[30:], g8[30:])             # ... and these are still queued when the pointer is asked for
    ptr = st.device_ptr
    assert _bitmap(st, n) == list(range(n))
    raw = (ctypes.c_double * (2 << n))()
    hip = ctypes.CDLL('libamdhip64.so.7')          # (the runtime the engine is linked against: already mapped)

    def read():
      st.sync()
      assert hip.hipMemcpy(raw, ctypes.c_void_p(ptr), ctypes.c_size_t(16 << n), 2) == 0   # DeviceToHost through the RAW pointer
      return np.frombuffer(raw, dtype=np.complex128).copy()

    assert np.max(np.abs(read() - want1)) < 1e-12
    st.run_stream(more_ops, more_g)              # scattered tiles again: would re-lay out if the handle still could
    st.flush()
    assert st.device_ptr == ptr and _bitmap(st, n) == list(range(n))
    assert np.max(np.abs(read() - want2)) < 1e-12
    assert np.max(np.abs(st.download() - want2)) < 1e-12
