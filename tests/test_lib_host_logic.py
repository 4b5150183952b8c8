"""CPU tests of the host-side API mirror (qcc_amd.lib): gate streams, register
bookkeeping, snapshot semantics, composite gates -- with the GPU replaced by an
oracle-backed stand-in through the backend test seams (tests/fake_device.py)."""
import math
import os

import numpy as np
import pytest

from qcc_amd.lib import backend, circuit, ops, state, tensor
from tests import fake_device


@pytest.fixture(autouse=True)
def cpu_backend():
  tensor.set_tensor_width(128)
  backend.set_device_factory(fake_device.OracleDevice)
  backend.set_host_mapped_factory(fake_device.OracleDevice)
  backend.set_host_executor(fake_device.OracleHostExecutor())
  yield
  backend.set_device_factory(None)
  backend.set_host_mapped_factory(None)
  backend.set_host_executor(None)
  tensor.set_tensor_width(None)


def _trace_of(qc):
  qc.flush()                 # eager gates queue on the host side until something reads the state (circuit.qc._drain)
  tr = qc._dev.trace
  ops_ = np.array([(-(2 ** 31) if c is None else c, t) for c, t, _ in tr], dtype=np.int64)
  gs = np.array([g for _, _, g in tr]).view(np.float64).reshape(-1, 8)
  return ops_, gs


def test_qft12_matches_reference_golden(golden_dir):
  g = np.load(os.path.join(golden_dir, 'g1_qft12.npz'))
  qc = circuit.qc('qft12')
  reg = qc.reg(12, tuple(int(b) for b in g['bits']))
  qc.qft(reg)
  ops_, gs = _trace_of(qc)
  assert np.array_equal(ops_, g['ops'].astype(np.int64))     # same native call stream
  assert np.array_equal(gs, g['gates'])                      # same gate doubles
  assert np.max(np.abs(qc.psi - g['final'])) < 1e-14


def test_multi_control_and_friends_match_reference_trace(golden_dir):
  g = np.load(os.path.join(golden_dir, 'g5_multi_control.npz'))
  qc = circuit.qc('mc')
  qc.reg(4, (1, 0, 1, 1))
  aux = qc.reg(4, 0)
  qc.h([0, 1, 2, 3])
  qc.multi_control([0, [1], 2], 3, aux, ops.PauliX(), 'mc-x')
  qc.multi_control([0, 1, [2], 3], 7, aux, ops.Hadamard(), 'mc-h')
  qc.cswap(0, 1, 2)
  qc.swap(0, 3)
  qc.ccu1(0, 1, 2, 0.77)
  qc.crx(1, 2, 0.3); qc.cry([0], 3, 0.4); qc.crz(3, 0, 0.5)
  ops_, gs = _trace_of(qc)
  assert np.array_equal(ops_, g['ops'].astype(np.int64))
  assert np.max(np.abs(gs - g['gates'])) < 1e-15             # sqrtm closed form vs scipy
  assert np.max(np.abs(qc.psi - g['final'])) < 1e-13


def test_qft_inverse_qft_traces(golden_dir):
  for n in (4, 7, 10):
    g = np.load(os.path.join(golden_dir, f'g5_qft_iqft_n{n}.npz'))
    qc = circuit.qc('qft')
    reg = qc.reg(n, (0b1011001110 >> (10 - n)))
    qc.qft(reg)
    qc.inverse_qft(list(reg)[: n // 2])
    assert np.array_equal(_trace_of(qc)[0], g['ops'].astype(np.int64))
    assert np.max(np.abs(qc.psi - g['final'])) < 1e-14


def test_state_apply_methods_and_negative_controls(golden_dir):
  g = np.load(os.path.join(golden_dir, 'py_fallback.npz'))
  psi = state.bitstring(1, 0, 1, 0, 1)
  for (c, t), gate in zip(g['ops'], g['gates'].view(np.complex128).reshape(-1, 2, 2)):
    if c == -(2 ** 31):
      psi.apply1(ops.Operator(gate), int(t))
    else:
      psi.applyc(ops.Operator(gate), int(c), int(t))
  assert np.max(np.abs(psi - g['final'])) < 1e-14


def test_registers_stay_symbolic_until_needed():
  qc = circuit.qc('lazy')
  qc.reg(20, 5)
  qc.reg(1, 1)
  qc.bitstring(1, 0)
  qc.zeros(3); qc.ones(2)
  assert qc.nbits == 28 and qc._host is None and qc._dev is None   # nothing materialised
  from qcc_amd.device import merge_factors
  assert merge_factors(qc._factors) == [(28, (((((5 << 1) | 1) << 2 | 0b10) << 3) << 2) | 0b11)]
  # non-basis pieces stay factors too (built by qh_init_product on first use, never np.kron'ed)
  qc.qubit(0.6, 0.8)
  qc.reg(3, 1)
  assert qc.nbits == 32 and qc._host is None and qc._dev is None
  m = merge_factors(qc._factors)
  assert [f[0] for f in m] == [28, 4] and np.allclose(m[1][1][[1, 9]], [0.6, 0.8])
  qc2 = circuit.qc('small')
  qc2.reg(3, 5)
  assert np.array_equal(np.asarray(qc2.psi), np.asarray(state.bitstring(1, 0, 1)))
  qc3 = circuit.qc('mixed')
  qc3.reg(2, 2)
  qc3.qubit(0.6, 0.8)
  qc3.h(0)                                     # first gate: product built by the (fake) device
  want = state.bitstring(1, 0) * state.qubit(0.6, 0.8)
  want.apply1(ops.Hadamard(), 0)
  assert np.allclose(np.asarray(qc3.psi), np.asarray(want))


def test_snapshot_semantics():
  qc = circuit.qc('snap')
  qc.reg(3, 0)
  qc.h(0)
  before = qc.psi
  assert not before.flags.writeable
  assert qc.psi is before                      # no new download without new gates
  qc.x(2)
  after = qc.psi
  assert after is not before and abs(before[0]) > 0.7 and abs(after[1]) > 0.7
  qc.psi = before                              # entanglement_swap.py:80-83 pattern
  assert np.allclose(qc.psi, before)
  qc.h(0)
  assert np.allclose(np.abs(qc.psi[0]), 1.0)


def test_measure_bit_matches_projector_formulation():
  qc = circuit.qc('m')
  qc.reg(4, 0)
  qc.h(0); qc.cx(0, 1); qc.ry(2, 0.7); qc.cu1(2, 3, 0.4); qc.h(3)
  ref = qc.psi
  for idx in range(4):
    for to in (0, 1):
      p_host, collapsed_host = ops.Measure(ref, idx, to, True)
      qc.psi = ref
      p, snap = qc.measure_bit(idx, to, True)
      assert abs(p - p_host) < 1e-14
      assert np.allclose(snap, collapsed_host, atol=1e-14)
  qc.psi = ref
  assert abs(qc.pauli_expectation(0)) < 1e-14
  bits, p = qc.maxprob()
  hb, hp = ref.maxprob()
  assert bits == hb and abs(p - hp) < 1e-15


def test_measure_bit_returns_a_real_state_snapshot():
  """circuit.py:291-297 hands back (prob, State): dispatch on isinstance(.., State) and arithmetic must work,
  and the value must be the state at measurement time, not at read time."""
  qc = circuit.qc('m')
  qc.reg(3, 0)
  qc.h(0); qc.cx(0, 1)
  p, psi = qc.measure_bit(0, 1, collapse=True)
  assert isinstance(psi, state.State) and abs(p - 0.5) < 1e-14
  kept = np.array(psi)
  qc.h(2)                                         # more gates: the snapshot must not follow them
  assert np.array_equal(np.array(psi), kept)
  assert abs(np.vdot(psi, psi) - 1) < 1e-14 and np.allclose((psi * 2)[6], 2 * kept[6])
  # large registers get the lazy handle: same protocol, snapshot on first look
  old = circuit._MEASURE_SNAPSHOT_BITS
  circuit._MEASURE_SNAPSHOT_BITS = 2
  try:
    p2, lazy = qc.measure_bit(1, 1, collapse=False)
    assert isinstance(lazy, circuit._LazyPsi) and abs(p2 - 1.0) < 1e-14
    first = np.array(lazy)                        # first look: snapshot
    qc.x(2)
    assert np.array_equal(np.array(lazy), first)
    assert np.allclose(lazy * 2, 2 * first) and np.allclose(2 * lazy, 2 * first) and np.allclose(np.abs(lazy), np.abs(first))
    assert abs((lazy.conj() @ lazy) - 1) < 1e-13 and lazy.nbits == 3 and (lazy == first).all()
  finally:
    circuit._MEASURE_SNAPSHOT_BITS = old


def test_ir_inverse_control_by_and_run():
  main = circuit.qc('main')
  main.reg(4, 0b0110)
  start = main.psi
  sub = main.sub('blk')
  sub.h(0); sub.cu1(0, 1, 0.3); sub.rx(1, 0.2); sub.cx(1, 2)
  assert not sub.eager and sub.ir.ngates == 4
  main.qc(sub, offset=1)
  main.qc(sub.inverse(), offset=1)
  assert np.allclose(main.psi, start, atol=1e-14)
  c = circuit.qc('c', eager=False)
  c.h(1); c.cx(1, 2)
  c.control_by(0)
  names = [str(n) for n in c.ir.gates if n.is_gate()]
  assert names[0] == 'ch(0, 1)' and len(names) == 6       # h -> ch ; cx -> 5-gate ccx
  lazy = circuit.qc('lazy', eager=False)
  lazy.reg(2, 0)
  lazy.h(0); lazy.cx(0, 1)
  assert lazy._dev is None
  lazy.run()
  assert np.allclose(np.abs(lazy.psi) ** 2, [0.5, 0, 0, 0.5])
  assert 'Gates : 2' in lazy.stats()


def test_operator_and_tensor_api():
  x = ops.PauliX()
  assert (ops.Hadamard() @ ops.Hadamard()).is_close(ops.Identity())
  assert ops.Cnot(0, 1)(state.bitstring(1, 0)).is_close(state.bitstring(1, 1))
  assert ops.Swap(0, 2)(state.bitstring(1, 0, 0)).is_close(state.bitstring(0, 0, 1))
  assert ops.Toffoli(0, 1, 2)(state.bitstring(1, 1, 0)).is_close(state.bitstring(1, 1, 1))
  assert ops.Cnot0(0, 1)(state.bitstring(0, 0)).is_close(state.bitstring(0, 1))
  assert x(state.zeros(3), 1).is_close(state.bitstring(0, 1, 0))
  assert ops.Rk(2).is_close(ops.Sgate()) and ops.U3(0, 0, 0).is_unitary()
  assert (x * x).nbits == 2 and x.kpow(0).shape == ()
  assert ops.Vgate().is_unitary() and ops.PauliZ().is_hermitian() and x.is_permutation()
  assert state.plus(2).is_close([0.5] * 4) and abs(state.qubit(alpha=0.6)[1] - 0.8) < 1e-7
  assert str(state.Reg(3, 5, 2)) == '|101>' and list(state.Reg(3, 5, 2)) == [2, 3, 4]


def test_complex64_policy():
  tensor.set_tensor_width(64)
  qc = circuit.qc('w64')
  qc.reg(3, 1)
  qc.h(0); qc.cx(0, 2)
  assert qc.psi.dtype == np.complex64 and qc._dev.bit_width == 64
  assert np.allclose(np.abs(qc.psi) ** 2, [0, 0.5, 0, 0, 0.5, 0, 0, 0], atol=1e-6)


def test_width_change_under_a_live_state_keeps_the_amplitudes():
  """The reference silently stops applying gates when tensor_width no longer matches psi's dtype (SURVEY appendix B,
  Q1); here the state is carried over to the new width, with gates still queued on the host side."""
  tensor.set_tensor_width(128)
  try:
    qc = circuit.qc('w')
    qc.reg(3, 1)
    qc.h(0); qc.cx(0, 2); qc.flush()
    qc.h(1)                                  # queued on the host side at width 128
    tensor.set_tensor_width(64)
    qc.x(2)
    psi = np.asarray(qc.psi)
    assert psi.dtype == np.complex64 and qc._dev.bit_width == 64
    want = np.zeros(8); want[[0, 2, 5, 7]] = 0.25
    assert np.allclose(np.abs(psi) ** 2, want, atol=1e-6)
  finally:
    tensor.set_tensor_width(128)


def alias_contract(make_qc, oracle_apply):
  """The reference's in-place contract (src/lib/xgates.cc:37-38; relied on by code that keeps `psi`
  around): shared by the CPU stand-in test below and tests/test_gpu_lib.py on the real host-mapped state."""
  qc = make_qc()
  qc.reg(3, (1, 0, 1))
  qc.qubit(0.6, 0.8)
  qc.h(0)
  p = qc.psi
  assert p.flags.writeable and qc.psi is p             # one State over one buffer
  before = np.array(p)
  qc.h(1)
  qc.cx(0, 3)
  want = before.copy()
  oracle_apply(want, 4, [(None, 1, ops.Hadamard()), (0, 3, ops.PauliX())])
  assert np.max(np.abs(p - want)) < 1e-12               # the holder of `p` sees the gates
  # a slice taken earlier aliases too (grover.py:164-style readers)
  head = p[:4]
  qc.z(3)
  oracle_apply(want, 4, [(None, 3, ops.PauliZ())])
  assert np.max(np.abs(head - want[:4])) < 1e-12
  # writes through the view are the state
  p[:] = 0
  p[5] = 1
  qc.x(3)
  assert abs(qc.psi[4] - 1) < 1e-12 and abs(np.vdot(p, p).real - 1) < 1e-12
  # assignment of a same-size state: the buffer stays THE state
  other = state.bitstring(1, 1, 0, 0)
  qc.psi = other
  assert qc.psi is p and abs(p[12] - 1) < 1e-12
  # measurement collapses in place
  qc.h(0)
  prob, _ = qc.measure_bit(0, 1, collapse=True)
  assert abs(prob - 0.5) < 1e-12 and abs(abs(p[12]) - 1) < 1e-12 and abs(p[4]) < 1e-12
  # growing the register makes a new buffer (as `self.psi = self.psi * new` does in circuit.py:121-123);
  # the old view stays valid memory
  old = np.array(p)
  qc.reg(2, 0)
  assert qc.psi.nbits == 6 and qc.psi is not p
  assert np.array_equal(np.array(p), old)
  qc.h(5)
  assert np.array_equal(np.array(p), old)


def test_alias_psi_contract_on_the_stand_in(oracle):
  def apply(psi, n, gl):
    for c, t, g in gl:
      if c is None:
        oracle.apply1(psi, np.asarray(g).reshape(4), n, t)
      else:
        oracle.applyc(psi, np.asarray(g).reshape(4), n, c, t)
  alias_contract(lambda: circuit.qc('alias', alias_psi=True), apply)
  # the default stays a read-only snapshot
  qc = circuit.qc('snap')
  qc.reg(3, 0)
  qc.h(0)
  s0 = qc.psi
  qc.h(1)
  assert not s0.flags.writeable and abs(s0[2]) < 1e-15 and abs(qc.psi[2]) > 0.1


def test_device_pool_survives_the_cycle_collector(monkeypatch):
  """qcc_amd.lib.backend parks the device state of a finished circuit for the next one of the same shape.  A circuit
  holds reference cycles (its gate lambdas), so it is usually reclaimed by the cycle collector -- which runs the
  finalizer of the circuit (parks the state, resurrecting it) AND the finalizer of the state in the same pass: the
  latter must leave a parked handle alone (round 4: it closed it, the next circuit got a dead handle)."""
  import gc
  from qcc_amd import device, native

  class FakeLib:
    def qh_discard_pending(self, h):
      return 0

    def qh_destroy(self, h):
      return 0

  class DeviceState(device.DeviceState):            # same finalizer / close() as the real one, no device behind it
    created = 0

    def __init__(self, nbits, bw):
      DeviceState.created += 1
      self.nbits, self.bit_width = nbits, bw
      self.h, self.lib = object(), FakeLib()

    def init_product(self, factors):
      assert self.h is not None, 'a closed handle came out of the pool'

    def run_stream(self, ops_, g):
      pass

    def argmax(self):
      return 0, 1.0

    def reset_stats(self):
      pass

  monkeypatch.setattr(native, 'check', lambda rc: None)
  backend.set_device_factory(None)                  # the pool only serves the default factory
  monkeypatch.setattr(backend, '_default_device_factory', lambda n, bw: DeviceState(n, bw))
  backend.drop_device_pool()
  qc1 = circuit.qc('first')
  r = qc1.reg(10, 5)
  qc1.qft(r)
  qc1.maxprob()
  first = qc1._dev
  # the generated gate methods hold the circuit weakly (no reference cycle): dropping the last name parks the state AT
  # ONCE -- a circuit waiting for the cycle collector kept two 16-GiB buffers and the next circuit allocated afresh
  gc.disable()
  try:
    del qc1, r
    assert backend._pool_order == [first] and first.h is not None, 'the circuit was not released by reference counting'
  finally:
    gc.enable()
  gc.collect()
  assert backend._pool_order == [first] and first.h is not None
  qc2 = circuit.qc('second')
  r = qc2.reg(10, 11)
  qc2.h(r[9])
  qc2.maxprob()
  assert qc2._dev is first and DeviceState.created == 1 and not backend._pool_order
  qc3 = circuit.qc('third')                         # pool empty: a new state
  qc3.reg(10, 1)
  qc3.h(0)
  qc3.maxprob()
  assert DeviceState.created == 2
  qc2.close(); qc3.close()
  assert len(backend._pool_order) == 2
  monkeypatch.setenv('QCC_POOL_STATES', '1')
  qc4 = circuit.qc('fourth')
  qc4.reg(9, 0); qc4.h(0); qc4.maxprob()
  qc4.close()                                       # a third parked state: the oldest ones are closed
  assert len(backend._pool_order) == 1 and backend._pool_order[0].nbits == 9
  backend.drop_device_pool()
  assert not backend._pool_order and first.h is None


def test_state_mirror_keeps_the_host_contract(monkeypatch):
  """State.apply1 / applyc keep a device mirror (qcc_amd/lib/state.py): one upload, gates on the device, one download the
  first time anything looks.  With the oracle stand-in as the device: whatever the caller does between the gates -- index,
  slice, iterate, print, NumPy functions and operators, attributes that alias the buffer, copies, pickling, writes -- it
  sees, and leaves, exactly what the literal per-call path (the reference's contract, state.py:80-125) would."""
  import copy
  import pickle
  monkeypatch.setenv('QCC_STATE_MIRROR_MIN_QUBITS', '3')
  rng = np.random.default_rng(5)
  n = 7
  looks = [
      lambda p: p[3], lambda p: p[2:9].copy(), lambda p: list(p)[5], lambda p: repr(p), lambda p: str(p),
      lambda p: np.abs(p).sum(), lambda p: np.vdot(p, p), lambda p: (p + 1)[0], lambda p: p @ p, lambda p: p.real[7],
      lambda p: p.imag.sum(), lambda p: p.conj()[1], lambda p: p.sum(), lambda p: p.tolist()[3], lambda p: p.copy()[4],
      lambda p: copy.deepcopy(p)[4], lambda p: pickle.loads(pickle.dumps(p))[6], lambda p: p.view(np.ndarray)[2],
      lambda p: p.reshape(2, -1)[1, 3], lambda p: np.linalg.norm(p), lambda p: p.prob(0, 1, 0, 1, 0, 1, 1),
      lambda p: p.maxprob(), lambda p: (p * state.ones(1))[9], lambda p: p.density()[2, 3], lambda p: p.tobytes()[:16],
      lambda p: bytes(memoryview(p.data))[:16], lambda p: p.ctypes.data, lambda p: p.astype(np.complex64)[1],
      lambda p: np.allclose(p, p), lambda p: p.normalize()[0], lambda p: float(np.real(p[1])), lambda p: 5 in p,
      lambda p: np.concatenate([p, p])[130], lambda p: np.kron(p, [1, 0])[4], lambda p: p.ampl(1, 0, 1, 0, 1, 0, 1),
  ]

  def run(mirror):
    backend.set_state_mirror(mirror)
    state.mirror_stats(reset=True)
    r = np.random.default_rng(77)
    v = r.standard_normal(1 << n) + 1j * r.standard_normal(1 << n)
    psi = state.State(v / np.linalg.norm(v))
    seen = []
    for step in range(3 * len(looks)):
      for _ in range(int(r.integers(1, 6))):
        g = ops.Operator((r.standard_normal((2, 2)) + 1j * r.standard_normal((2, 2))) / 1.6)
        if r.random() < 0.5:
          psi.apply1(g, int(r.integers(0, n)))
        else:
          c, t = (int(x) for x in r.choice(n, 2, replace=False))
          psi.applyc(g, c, t)
      out = looks[step % len(looks)](psi)
      seen.append(np.asarray(out, dtype=object if isinstance(out, (str, bytes, tuple)) else None))
      if step % 7 == 3:                     # the caller writes to the array between gates
        psi[int(r.integers(0, 1 << n))] = 0.25
      if step % 11 == 5:
        psi *= 0.5
    return np.array(psi), seen, state.mirror_stats()

  try:
    want, seen_w, st_w = run(False)
    got, seen_g, st_g = run(True)
  finally:
    backend.set_state_mirror(None)
  assert st_w['gates'] == 0 and st_g['gates'] > 250 and st_g['uploads'] == st_g['downloads'] == 3 * len(looks)
  assert np.allclose(got, want, atol=1e-12)
  for a, b in zip(seen_g, seen_w):
    if a.dtype == object:
      assert a.shape == b.shape and (a.tolist() == b.tolist() or isinstance(a.tolist(), int))   # (ctypes.data: an address)
    else:
      assert np.allclose(a.astype(np.complex128), b.astype(np.complex128), atol=1e-10)


def test_state_mirror_moves_the_state_twice_for_a_run_of_gates(monkeypatch):
  """40 direct apply calls = one upload + one download, views and copies never share a mirror, a State
  that owns its buffer and dies with gates pending downloads nothing, one that aliases somebody's array brings them home."""
  monkeypatch.setenv('QCC_STATE_MIRROR_MIN_QUBITS', '3')
  backend.set_state_mirror(True)
  try:
    state.mirror_stats(reset=True)
    psi = state.zeros(8)
    ref = np.zeros(1 << 8, dtype=np.complex128)
    ref[0] = 1
    from tests import oracle_lib
    o = oracle_lib.load()
    for k in range(40):
      psi.apply1(ops.Hadamard(), k % 8)
      o.apply1(ref, np.asarray(ops.Hadamard()).reshape(4), 8, k % 8)
      if k % 3 == 0:
        psi.applyc(ops.PauliX(), k % 8, (k + 3) % 8)
        o.applyc(ref, np.asarray(ops.PauliX()).reshape(4), 8, k % 8, (k + 3) % 8)
    st = state.mirror_stats()
    assert st['uploads'] == 1 and st['downloads'] == 0 and st['h2d_bytes'] == 16 << 8
    assert np.allclose(np.asarray(psi[:]), ref, atol=1e-12)          # the first look: one download
    st = state.mirror_stats()
    assert st['downloads'] == 1 and st['d2h_bytes'] == 16 << 8 and st['gates'] == 54
    v = psi[4:20]                                                    # a view has no mirror of its own
    assert v._mirror is None
    # ... but it shares the memory: a view taken BEFORE the gates sees them (the reference's arrays alias), whichever object
    # holds the mirror -- and the array a view was taken from sees the gates applied THROUGH the view
    before = np.array(v)
    psi.apply1(ops.PauliX(), 7)                                      # (qubit 7 = index bit 0: swaps neighbours)
    o.apply1(ref, np.asarray(ops.PauliX()).reshape(4), 8, 7)
    assert np.isnan(np.asarray(v)).all()                             # what bypasses the hooks reads poison, not a stale state
    assert np.allclose(v[:], ref[4:20], atol=1e-12) and not np.allclose(v[:], before)
    low = psi[0:128]                                                 # a 7-qubit State over the lower half
    low.apply1(ops.Hadamard(), 0)
    o.apply1(ref[0:128], np.asarray(ops.Hadamard()).reshape(4), 7, 0)
    assert np.allclose(psi[:], ref, atol=1e-12)
    low.apply1(ops.PauliX(), 2)
    o.apply1(ref[0:128], np.asarray(ops.PauliX()).reshape(4), 7, 2)
    del low                                                          # the view dies with a gate pending: its parent still sees it
    assert np.allclose(psi[:], ref, atol=1e-12)
    n_down = state.mirror_stats()['downloads']
    n_up = state.mirror_stats()['uploads']
    del v, before
    # a State made from an array ALIASES it (State(a) is a view of a): when it dies with a gate pending the gates go home,
    # unconditionally (no reference-count guessing) -- whoever holds `a` sees them
    psi.apply1(ops.Hadamard(), 0)
    o.apply1(ref, np.asarray(ops.Hadamard()).reshape(4), 8, 0)
    raw = psi.base
    assert raw is not None and type(raw) is np.ndarray
    del psi
    assert state.mirror_stats()['downloads'] == n_down + 1 and state.mirror_stats()['uploads'] == n_up + 1
    assert np.allclose(raw, ref, atol=1e-12)
    # only a State that OWNS its buffer (no base: nobody else can reach the memory once it is gone) dies without a download
    own = state.State(ref).copy()
    assert own.base is None
    own.apply1(ops.Hadamard(), 3)
    del own
    assert state.mirror_stats()['downloads'] == n_down + 1 and state.mirror_stats()['uploads'] == n_up + 2
  finally:
    backend.set_state_mirror(None)


def test_state_mirror_is_opt_in_and_costs_nothing_when_off(monkeypatch):
  """the device mirror is OFF unless the caller opts in -- the default keeps the reference's literal contract
  (the host array holds the result when apply1 returns: state.py:80-125) --, a State carries no Python-level hooks until
  the first mirror is made and loses them again when the mode is off and the last mirror has gone home, leaving a
  `with state.device_mirror()` block brings every mirror home, a user subclass of State goes through the ufunc hook
  without recursing, and the byte range of an array does not need NumPy >= 2."""
  monkeypatch.setenv('QCC_STATE_MIRROR_MIN_QUBITS', '3')
  monkeypatch.delenv('QCC_STATE_MIRROR', raising=False)
  assert backend.state_mirror_setting() is None and not backend.state_mirror_allowed()
  hooks = ('__getattribute__', '__getitem__', '__array_ufunc__', '__array_function__', '__del__', '__repr__')
  assert not any(h in state.State.__dict__ for h in hooks)
  state.mirror_stats(reset=True)
  a = np.zeros(1 << 6, dtype=np.complex128)
  a[0] = 1
  psi = state.State(a)                       # aliases a
  psi.apply1(ops.Hadamard(), 0)
  assert state.mirror_stats()['gates'] == 0 and abs(a[0] - 2 ** -0.5) < 1e-12 and abs(a[32] - 2 ** -0.5) < 1e-12
  assert not any(h in state.State.__dict__ for h in hooks)

  class MyState(state.State):
    pass

  with state.device_mirror():
    assert backend.state_mirror_allowed()
    psi.apply1(ops.Hadamard(), 1)
    assert all(h in state.State.__dict__ for h in hooks)
    assert state.mirror_stats()['gates'] == 1 and np.isnan(a).all()       # raw memory: poison (silent), documented
    mine = np.ndarray.view(state.State(np.ones(1 << 6, dtype=np.complex128)), MyState)
    mine.apply1(ops.PauliX(), 2)
    assert type(mine + mine) is MyState and np.allclose(np.asarray((mine + mine)[:]), 2.0)     # no recursion
    assert type(np.add(mine, 1.0, out=mine)) is MyState
  # the block has ended: everything is home, the raw array the State was made from holds the result, hooks are gone
  assert not backend.state_mirror_allowed() and not state._LIVE
  assert abs(a[0] - 0.5) < 1e-12 and abs(a[16] - 0.5) < 1e-12 and abs(a[32] - 0.5) < 1e-12 and abs(a[48] - 0.5) < 1e-12
  assert not any(h in state.State.__dict__ for h in hooks)
  assert state.mirror_stats()['downloads'] >= 1
  # QCC_STATE_MIRROR=1 opts in from the environment -- but never while a test executor is installed
  monkeypatch.setenv('QCC_STATE_MIRROR', '1')
  assert not backend.state_mirror_allowed()
  # byte range without numpy.lib.array_utils (NumPy 1.x without byte_bounds either)
  monkeypatch.setattr(state, '_byte_bounds', None)
  x = np.arange(40, dtype=np.complex128)
  lo = x.__array_interface__['data'][0]
  assert state._byte_range(x) == (lo, lo + 640)
  assert state._byte_range(x[4:20:3]) == (lo + 64, lo + 64 + 15 * 16 + 16)
  assert state._byte_range(x[::-1]) == (lo, lo + 640)
