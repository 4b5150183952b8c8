"""GPU tests of the host API mirror on the real engine (no stand-ins): the
reference's circuits re-issued through qcc_amd.lib.circuit.qc run on the MI355X
and reproduce the reference's golden final states."""
import importlib
import math
import os
import sys

import numpy as np
import pytest

from qcc_amd import native, workloads
from qcc_amd.lib import backend, circuit, ops, state, tensor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def width128():
  backend.set_device_factory(None)
  backend.set_host_executor(None)
  tensor.set_tensor_width(128)
  yield
  tensor.set_tensor_width(None)


def test_qc_qft12_config1(golden_dir):
  g = np.load(os.path.join(golden_dir, 'g1_qft12.npz'))
  qc = circuit.qc('qft12')
  reg = qc.reg(12, tuple(int(b) for b in g['bits']))
  qc.qft(reg)
  assert np.max(np.abs(qc.psi - g['final'])) <= 1e-10       # BASELINE config 1, vs reference xgates
  g2 = np.load(os.path.join(golden_dir, 'g2_libq_qft12.npz'))  # and vs reference libq (float)
  dense = np.zeros(1 << 12, dtype=np.complex128)
  for s, a in zip(g2['libq_state'], g2['amp']):
    dense[int(format(int(s), '012b')[::-1], 2)] = a
  assert np.max(np.abs(qc.psi - dense)) < 2e-6
  assert type(qc._dev).__module__ == 'qcc_amd.device'       # really the HIP engine


def test_qc_composites_match_reference(golden_dir):
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
  assert np.max(np.abs(qc.psi - g['final'])) <= 1e-12


def test_grover_circuit_reference_and_recurrence(golden_dir):
  """grover.py:124-168 through qc (nbits=6 -> 12 qubits): reference golden state,
  the 4-amplitude recurrence and the device-side maxprob."""
  g = np.load(os.path.join(golden_dir, 'g5_grover6.npz'))
  nb, bits = 6, [int(b) for b in g['marked']]
  qc = circuit.qc('Grover')
  reg = qc.reg(nb, 0)
  qc.reg(1, 1)
  aux = qc.reg(nb - 1, 0)
  its = int(math.pi / 4 * math.sqrt(2 ** nb))
  qc.h(list(range(nb + 1)))
  idx = list(range(nb))
  for _ in range(its):
    for i in idx:
      if bits[i] == 0:
        qc.apply1(ops.PauliX(), i, 'x')
    qc.multi_control(reg, nb, aux, ops.PauliX(), 'Phase Inversion')
    for i in idx:
      if bits[i] == 0:
        qc.apply1(ops.PauliX(), i, 'x')
    qc.h(idx); qc.x(idx)
    qc.multi_control(reg, nb, aux, ops.PauliZ(), 'Mean Inversion')
    qc.x(idx); qc.h(idx)
  maxbits, maxprob = qc.maxprob()                     # device-side argmax
  assert maxbits[:nb] == bits
  psi = qc.psi
  assert np.max(np.abs(psi - g['final'])) <= 1e-10
  cm0, cm1, cu0, cu1 = workloads.grover_recurrence(nb, its)
  x = int(''.join(map(str, bits)), 2)
  other = (x + 1) % (1 << nb)
  for xx, (c0, c1) in ((x, (cm0, cm1)), (other, (cu0, cu1))):
    assert abs(psi[(xx << nb) | 0] - c0) < 1e-12 and abs(psi[(xx << nb) | (1 << (nb - 1))] - c1) < 1e-12
  assert abs(maxprob - max(cm0 ** 2, cm1 ** 2)) < 1e-12


def test_state_methods_run_on_gpu(golden_dir):
  g = np.load(os.path.join(golden_dir, 'py_fallback.npz'))
  psi = state.bitstring(1, 0, 1, 0, 1)
  for (c, t), gate in zip(g['ops'], g['gates'].view(np.complex128).reshape(-1, 2, 2)):
    if c == -(2 ** 31):
      psi.apply1(ops.Operator(gate), int(t))
    else:
      psi.applyc(ops.Operator(gate), int(c), int(t))      # includes the negative-control calls
  assert np.max(np.abs(psi - g['final'])) < 1e-13


def test_state_methods_through_the_device_mirror(golden_dir, oracle, monkeypatch):
  """Direct State.apply1 / applyc calls keep the state on the device between gates (qcc_amd/lib/state.py: one upload, fused
  sweeps, one download at the first look) -- VERDICT r04 #6: 40 direct calls on a 24-qubit State move <= 2 x S over PCIe,
  the amplitudes equal the oracle's; the reference's own recorded fallback sequence (py_fallback.npz, negative controls
  included) gives the golden result through the mirror as well."""
  from qcc_amd.lib import backend
  assert not backend.state_mirror_allowed()               # opt-in: the default is the literal per-call path
  backend.set_state_mirror(True)
  try:
    _state_methods_through_the_device_mirror(golden_dir, oracle, monkeypatch)
  finally:
    backend.set_state_mirror(None)


def _state_methods_through_the_device_mirror(golden_dir, oracle, monkeypatch):
  state.mirror_stats(reset=True)
  n = 24
  rng = np.random.default_rng(11)
  v = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
  v /= np.linalg.norm(v)
  want = v.copy()
  psi = state.State(v)
  for k in range(40):
    g = np.linalg.qr(rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)))[0]
    if k % 3:
      t = int(rng.integers(0, n))
      psi.apply1(ops.Operator(g), t)
      oracle.apply1(want, g.reshape(4), n, t)
    else:
      c, t = (int(x) for x in rng.choice(n, 2, replace=False))
      psi.applyc(ops.Operator(g), c, t)
      oracle.applyc(want, g.reshape(4), n, c, t)
  st = state.mirror_stats()
  assert st['gates'] == 40 and st['uploads'] == 1 and st['downloads'] == 0 and st['h2d_bytes'] == 16 << n
  err = float(np.max(np.abs(np.asarray(psi[:]) - want)))
  st = state.mirror_stats()
  assert st['downloads'] == 1 and st['h2d_bytes'] + st['d2h_bytes'] == 2 * (16 << n)      # <= 2 x S over PCIe
  assert err < 1e-12
  monkeypatch.setenv('QCC_STATE_MIRROR_MIN_QUBITS', '1')
  g = np.load(os.path.join(golden_dir, 'py_fallback.npz'))
  psi = state.bitstring(1, 0, 1, 0, 1)
  for (c, t), gate in zip(g['ops'], g['gates'].view(np.complex128).reshape(-1, 2, 2)):
    if c == -(2 ** 31):
      psi.apply1(ops.Operator(gate), int(t))
    else:
      psi.applyc(ops.Operator(gate), int(c), int(t))      # (the negative-control calls take the literal path)
  assert state.mirror_stats()['gates'] > 40
  assert np.max(np.abs(psi - g['final'])) < 1e-13


def test_libxgates_dropin_module(oracle):
  sys.path.insert(0, os.path.join(ROOT, 'qcc_amd', 'dropin'))
  try:
    xg = importlib.import_module('libxgates')
  finally:
    sys.path.pop(0)
  rng = np.random.default_rng(2)
  for bw, dt, tol in ((128, np.complex128, 1e-13), (64, np.complex64, 1e-6)):
    n = 7
    psi = (rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)).astype(dt)
    want = psi.copy()
    gate = (rng.standard_normal(4) + 1j * rng.standard_normal(4)).astype(dt)
    assert xg.apply1(psi, gate, n, 2, bw) is None
    oracle.apply1(want, gate, n, 2)
    xg.applyc(psi, gate, n, 6, 0, bw)
    oracle.applyc(want, gate, n, 6, 0)
    assert np.max(np.abs(psi - want)) < tol * 10
  with pytest.raises(TypeError):
    xg.apply1(np.zeros(4, dtype=np.complex64), np.eye(2).reshape(4), 2, 0, 128)


def test_measure_and_snapshots_on_gpu():
  qc = circuit.qc('m')
  qc.reg(5, 0b10110)
  qc.h(0); qc.cx(0, 1); qc.ry(2, 0.7); qc.cu1(2, 3, 0.4); qc.h(4)
  ref = qc.psi
  for idx in range(5):
    for to in (0, 1):
      p_host, collapsed = ops.Measure(ref, idx, to, True) if ops.Measure(ref, idx, to, False)[0] > 1e-9 else (0, None)
      if collapsed is None:
        continue
      qc.psi = ref
      p, snap = qc.measure_bit(idx, to, True)
      assert abs(p - p_host) < 1e-13 and np.allclose(snap, collapsed, atol=1e-13)
  qc.psi = ref
  qc.h(0)
  assert qc.psi is not ref and not qc.psi.flags.writeable


def test_large_register_never_touches_host():
  """28 qubits: reg() + gates + device-side readers, no 2^n host array (the
  reference materialises 2^n amplitudes per reg(): circuit.py:121-129)."""
  qc = circuit.qc('big')
  r = qc.reg(27, 0)
  qc.reg(1, 1)
  qc.h(0); qc.cx(0, 27 - 1); qc.x(5)
  assert qc._host is None
  bits, p = qc.maxprob()
  assert abs(p - 0.5) < 1e-12 and bits[5] == 1 and bits[27] == 1
  assert abs(qc.norm2() - 1) < 1e-12
  want0 = [0] * 28
  want0[5] = 1; want0[27] = 1
  assert abs(qc.prob(*want0) - 0.5) < 1e-12
  assert qc._host is None
  qc.close()


def test_product_registers_are_built_on_the_device():
  """qc.qubit(alpha, beta) / qc.state() next to large registers: no host kron (N4)."""
  from qcc_amd.lib import circuit, state
  qc = circuit.qc('product')
  qc.qubit(0.6, 0.8)
  qc.reg(27, 3)
  st = state.State(np.array([0.5, 0.5j, -0.5, 0.5]))
  qc.state(st)
  assert qc.nbits == 30 and qc._host is None
  qc.h(29)                                   # first gate -> qh_init_product on the device
  assert qc._host is None
  s = 1 / np.sqrt(2)
  # index = q<<29 | 3<<2 | k ;  H on the last qubit mixes k=0,1 and k=2,3
  assert abs(qc.ampl(*([0] + [0] * 25 + [1, 1] + [0, 0])) - 0.6 * s * (0.5 + 0.5j)) < 1e-12
  assert abs(qc.ampl(*([1] + [0] * 25 + [1, 1] + [1, 1])) - 0.8 * s * (-0.5 - 0.5)) < 1e-12
  assert abs(qc.norm2() - 1.0) < 1e-12
  qc.close()


def test_non_eager_circuit_runs_as_planned_sweeps():
  """a recorded (non-eager) circuit reaches the planner whole at qc.run() /
  qc.qc(sub) (ir.py, circuit.py:394-423) -- a 22-qubit QFT becomes a handful of sweeps."""
  n, x = 22, 0x2B5C3
  qc = circuit.qc('recorded', eager=False)
  reg = qc.reg(n, tuple(int(b) for b in format(x, f'0{n}b')))
  qc.qft(reg)
  assert qc.ir.ngates == n * (n + 1) // 2 and qc._dev is None       # nothing executed yet
  qc.run()
  qc.flush()                                     # (gates queue on the host side until something reads the state)
  dev = qc._dev
  st = dev.stats()
  assert st['gates_submitted'] == n * (n + 1) // 2 and st['kernels_launched'] <= 4 and st['sweeps'] == st['kernels_launched']
  ks = (0, 1, 12345, (1 << n) - 1)
  amps = workloads.qft_analytic(n, x, np.array(ks))
  got = np.array([qc.ampl(*[int(b) for b in format(int(k), f'0{n}b')]) for k in ks])
  assert np.max(np.abs(got - amps)) < 1e-12
  # the adjoint of the recorded circuit, replayed into an eager circuit holding QFT|x>
  main = circuit.qc('main')
  main.reg(n, tuple(int(b) for b in format(x, f'0{n}b')))
  main.qc(qc)
  main.qc(qc.inverse())
  bits, p = main.maxprob()
  assert abs(p - 1.0) < 1e-10 and helper_bits(bits) == x
  qc.close()
  main.close()


def helper_bits(bits):
  v = 0
  for b in bits:
    v = (v << 1) | int(b)
  return v


def test_alias_psi_contract_on_host_mapped_memory(oracle):
  """circuit.qc(alias_psi=True): the state lives in pinned host memory the GPU works on in place
  (qh_create_host_mapped); `p = qc.psi; qc.h(0)` changes `p` exactly as the reference's in-place
  xgates.cc:37-38 does.  Same checks as the CPU stand-in (tests/test_lib_host_logic.py)."""
  from tests.test_lib_host_logic import alias_contract

  def apply(psi, n, gl):
    for c, t, g in gl:
      if c is None:
        oracle.apply1(psi, np.asarray(g).reshape(4), n, t)
      else:
        oracle.applyc(psi, np.asarray(g).reshape(4), n, c, t)
  made = []

  def make():
    qc = circuit.qc('alias', alias_psi=True)
    made.append(qc)
    return qc
  alias_contract(make, apply)
  assert type(made[0]._dev).__module__ == 'qcc_amd.device'
  # a mid-size register through the same path: 20-qubit QFT, read through the alias only
  qc = circuit.qc('alias20', alias_psi=True)
  r = qc.reg(20, 0b1011)
  p = None
  qc.h(0)
  p = qc.psi
  qc.qft(r)
  ops_, g8 = workloads.qft_stream(range(20)).arrays()
  want = np.zeros(1 << 20, dtype=np.complex128)
  want[0b1011] = 1
  oracle.apply1(want, np.array([1, 1, 1, -1]) / np.sqrt(2), 20, 0)
  oracle.run_stream(want, 20, ops_, g8)
  assert np.max(np.abs(p - want)) < 1e-12


def test_device_states_are_reused_across_circuits():
  """qcc_amd.lib.backend's pool (reference call site src/lib/circuit.py:71-101: one qc per experiment):
  a circuit that is closed or collected parks its device state; the next circuit of the same shape gets that handle --
  buffers, second buffer and cached plans included -- and starts from ITS OWN initial state, not from what was left."""
  import gc
  gc.collect()                                     # (circuits of earlier tests park their states when they are collected)
  backend.drop_device_pool()
  n = 24
  qc1 = circuit.qc('first')
  r = qc1.reg(n, 5)
  qc1.qft(r)
  qc1.maxprob()
  h1 = qc1._dev.h.value
  del qc1                                          # collected -> parked
  gc.collect()
  assert sum(len(v) for v in backend._pool.values()) == 1
  qc2 = circuit.qc('second')
  r = qc2.reg(n, 0b1011)
  qc2.h(r[n - 1])
  bits, p = qc2.maxprob()
  assert qc2._dev.h.value == h1 and not backend._pool_order      # the same handle, taken out of the pool
  assert abs(p - 0.5) < 1e-12 and abs(qc2.norm2() - 1) < 1e-12
  assert abs(abs(qc2.ampl(*[int(b) for b in format(0b1011, f'0{n}b')])) ** 2 - 0.5) < 1e-12
  assert abs(abs(qc2.ampl(*[int(b) for b in format(0b1010, f'0{n}b')])) ** 2 - 0.5) < 1e-12
  qc3 = circuit.qc('other shape')                  # another shape: its own handle
  qc3.reg(20, 1)
  qc3.h(0)
  qc3.maxprob()
  assert qc3._dev.h.value != h1
  qc2.close(); qc3.close()
  assert len(backend._pool_order) == 2
  big = circuit.qc('too big to park')
  big.reg(28, 0); big.h(0); big.maxprob()
  os.environ['QCC_POOL_MAX_QUBITS'] = '27'
  try:
    big.close()
    assert len(backend._pool_order) == 2 and all(d.nbits != 28 for d in backend._pool_order)
  finally:
    del os.environ['QCC_POOL_MAX_QUBITS']
  backend.drop_device_pool()
  assert not backend._pool_order
