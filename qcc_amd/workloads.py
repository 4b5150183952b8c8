"""Gate streams of the reference workloads, as arrays the engine can replay.

Each generator reproduces the exact sequence of native apply1/applyc calls the
reference issues for that workload (checked against recorded traces in
tests/golden/g5_*.npz and g1_qft12.npz):

  qft_stream          qc.qft            src/lib/circuit.py:320-328
  inverse_qft_stream  qc.inverse_qft    src/lib/circuit.py:330-339
  supremacy_stream    build_circuit+sim_circuit  src/supremacy.py:123-158,208-253
  grover_stream       run_experiment_circuit     src/grover.py:124-168
                      (+ qc.multi_control / qc.ccu, circuit.py:341-392,227-246)

A stream is (ops int32[G,2], gates float64[G,8]); ops[k] = (ctl or NO_CTL, tgt)
in the reference's qubit numbering.
"""
import math
import random

import numpy as np

from qcc_amd import gates

NO_CTL = -(2 ** 31)


class StreamBuilder:
  """Collects native calls; mirrors the call surface of qc.apply1/applyc."""

  def __init__(self):
    self.ops, self.gs = [], []

  def apply1(self, gate, idx):
    self.ops.append((NO_CTL, int(idx)))
    self.gs.append(np.asarray(gate, dtype=np.complex128).reshape(4))

  def applyc(self, gate, ctl, idx):
    """ctl may be [q] = control-by-|0> (circuit.py:166-169,207-215): X, gate, X."""
    by0 = not isinstance(ctl, (int, np.integer))
    c = ctl[0] if by0 else ctl
    if by0:
      self.apply1(gates.pauli_x(), c)
    self.ops.append((int(c), int(idx)))
    self.gs.append(np.asarray(gate, dtype=np.complex128).reshape(4))
    if by0:
      self.apply1(gates.pauli_x(), c)

  def arrays(self):
    ops = np.array(self.ops, dtype=np.int32).reshape(-1, 2)
    g = np.array(self.gs, dtype=np.complex128).reshape(-1, 4)
    return ops, np.ascontiguousarray(g).view(np.float64).reshape(-1, 8)


def qft_stream(reg, sb=None):
  """reg: list of qubit numbers (circuit.py:320-328, no swaps)."""
  sb = sb or StreamBuilder()
  reg = list(reg)
  h = gates.hadamard()
  for i in reversed(range(len(reg))):
    sb.apply1(h, reg[i])
    for j in reversed(range(i)):
      sb.applyc(gates.u1(np.pi / 2 ** (i - j)), reg[i], reg[j])
  return sb


def inverse_qft_stream(reg, sb=None):
  sb = sb or StreamBuilder()
  reg = list(reg)
  h = gates.hadamard()
  for idx, r in enumerate(reg):
    sb.apply1(h, r)
    if idx != len(reg) - 1:
      for y in range(idx, -1, -1):
        sb.applyc(gates.u1(-np.pi / 2 ** (idx + 1 - y)), reg[idx + 1], reg[y])
  return sb


# The 8 CZ coupling patterns of the workload on a 6-wide qubit grid
# (data of the workload definition, supremacy.py:53-97): entry i != 0 couples
# qubit i with qubit i+entry (1 = right neighbour, 6 = the qubit below).
def _grid(rows):
  out = []
  for cols, off in rows:
    out += [off if c in cols else 0 for c in range(6)]
  return out


_E = ((), 0)
SUPREMACY_PATTERNS = [
    _grid([((2,), 1), ((0, 4), 1)] * 3),
    _grid([((0, 4), 1), ((2,), 1)] * 3),
    _grid([_E, ((1, 3, 5), 6), _E, ((1, 3, 5), 6), _E, _E]),
    _grid([_E, ((0, 2, 4), 6), _E, ((0, 2, 4), 6), _E, _E]),
    _grid([((3,), 1), ((1,), 1)] * 3),
    _grid([((1,), 1), ((3,), 1)] * 3),
    _grid([((0, 2, 4), 6), _E, ((1, 3, 5), 6), _E, ((0, 2, 4), 6), _E]),
    _grid([((1, 3, 5), 6), _E, ((1, 3, 5), 6), _E, ((1, 3, 5), 6), _E]),
]


def supremacy_layers(nbits, depth, patterns=None, rng=random):
  """supremacy.py:123-158 build_circuit: returns list of per-layer gate codes.

  Codes: 'h', 't', 'u', 'cz', None.  `patterns` is the reference's table of
  CZ offsets (list of 8 lists); consumes rng.randint(0,7) once per layer.
  """
  patterns = patterns or SUPREMACY_PATTERNS
  state0 = ['h'] * nbits
  states = [state0]
  for _ in range(depth - 1):
    state1 = [None] * nbits
    pat = patterns[rng.randint(0, 7)]
    for i in range(min(nbits, len(pat))):
      if pat[i] != 0 and i + pat[i] < nbits:
        state1[i] = 'cz'
        state1[i + pat[i]] = 'cz'
    for i in range(nbits):
      if state0[i] == 'cz' and state1[i] != 'cz':
        state1[i] = 'u'
      if state0[i] == 'u' and state1[i] != 'cz':
        state1[i] = 't'
      if state0[i] == 'h' and state1[i] != 'cz':
        state1[i] = 't'
    state0 = state1
    states.append(state0)
  states.append(['h'] * nbits)
  return states


def supremacy_stream(nbits, depth, seed=None, sb=None, patterns=None):
  """supremacy.py:208-253 sim_circuit gate stream (final H layer is NOT applied
  by the reference: the loop runs d in range(depth), quirk Q9)."""
  sb = sb or StreamBuilder()
  if seed is not None:
    random.seed(seed)
  states = supremacy_layers(nbits, depth, patterns)
  for d in range(depth):
    s = states[d]
    for i in range(nbits):
      if s[i] is None:
        continue
      if s[i] == 't':
        sb.apply1(gates.tgate(), i)
      if s[i] == 'h':
        sb.apply1(gates.hadamard(), i)
      if s[i] == 'u':
        if random.randint(0, 1) == 0:
          sb.apply1(gates.vgate(), i)
        else:
          sb.apply1(gates.yroot(), i)
      if s[i] == 'cz':
        if i < nbits - 1 and s[i + 1] == 'cz':
          sb.applyc(gates.pauli_z(), i, i + 1)
          s[i + 1] = None
        if i < nbits - 6 and s[i + 6] == 'cz':
          sb.applyc(gates.pauli_z(), i, i + 6)
          s[i + 6] = None
  return sb


def _sqrtm2(u):
  """Principal square root of a 2x2 matrix (what scipy.linalg.sqrtm returns
  for the unitaries used here, circuit.py:238), closed form:
  sqrt(M) = (M + s I) / t,  s = sqrt(det M), t = sqrt(tr M + 2 s)."""
  u = np.asarray(u, dtype=np.complex128)
  s = np.sqrt(u[0, 0] * u[1, 1] - u[0, 1] * u[1, 0])
  t = np.sqrt(u[0, 0] + u[1, 1] + 2 * s)
  if abs(t) < 1e-12:  # tr = -2s: take the other branch of s
    s = -s
    t = np.sqrt(u[0, 0] + u[1, 1] + 2 * s)
  return (u + s * np.eye(2)) / t


def ccu_stream(sb, idx0, idx1, idx2, op, sqrt_fn=_sqrtm2):
  """Sleator-Weinfurter controlled-controlled-U (circuit.py:227-246)."""
  def by0(c):
    return (c, False) if isinstance(c, (int, np.integer)) else (c[0], True)
  i0, c0 = by0(idx0)
  i1, c1 = by0(idx1)
  x = gates.pauli_x()
  if c0:
    sb.apply1(x, i0)
  if c1:
    sb.apply1(x, i1)
  v = np.asarray(sqrt_fn(op), dtype=np.complex128)
  sb.applyc(v, i0, idx2)
  sb.applyc(x, i0, i1)
  sb.applyc(gates.adjoint(v), i1, idx2)
  sb.applyc(x, i0, i1)
  sb.applyc(v, i1, idx2)
  if c1:
    sb.apply1(x, i1)
  if c0:
    sb.apply1(x, i0)


def multi_control_stream(sb, ctl, idx1, aux, gate, sqrt_fn=_sqrtm2):
  """qc.multi_control (circuit.py:341-392)."""
  ctl = list(ctl)
  if not ctl:
    sb.apply1(gate, idx1)
    return
  if len(ctl) == 1:
    sb.applyc(gate, ctl[0], idx1)
    return
  if len(ctl) == 2:
    ccu_stream(sb, ctl[0], ctl[1], idx1, gate, sqrt_fn)
    return
  x = gates.pauli_x()
  ccu_stream(sb, ctl[0], ctl[1], aux[0], x, sqrt_fn)
  a = 0
  for i in range(2, len(ctl)):
    ccu_stream(sb, ctl[i], aux[a], aux[a + 1], x, sqrt_fn)
    a += 1
  sb.applyc(gate, aux[a], idx1)
  a -= 1
  for i in range(len(ctl) - 1, 1, -1):
    ccu_stream(sb, ctl[i], aux[a], aux[a + 1], x, sqrt_fn)
    a -= 1
  ccu_stream(sb, ctl[0], ctl[1], aux[0], x, sqrt_fn)


def grover_stream(nbits, marked_bits, iterations=None, sb=None, sqrt_fn=_sqrtm2):
  """grover.py:124-168 run_experiment_circuit: 2*nbits qubits.

  Register layout: search reg = qubits 0..nbits-1, ancilla = nbits (init |1>),
  aux = nbits+1 .. 2*nbits-1.  Initial basis state has only the ancilla set.
  """
  sb = sb or StreamBuilder()
  reg = list(range(nbits))
  aux = list(range(nbits + 1, 2 * nbits))
  if iterations is None:
    iterations = int(math.pi / 4 * math.sqrt(2 ** nbits))
  h, x, z = gates.hadamard(), gates.pauli_x(), gates.pauli_z()
  for i in range(nbits + 1):
    sb.apply1(h, i)
  for _ in range(iterations):
    for i in reg:
      if marked_bits[i] == 0:
        sb.apply1(x, i)
    multi_control_stream(sb, reg, nbits, aux, x, sqrt_fn)
    for i in reg:
      if marked_bits[i] == 0:
        sb.apply1(x, i)
    for i in reg:
      sb.apply1(h, i)
    for i in reg:
      sb.apply1(x, i)
    multi_control_stream(sb, reg, nbits, aux, z, sqrt_fn)
    for i in reg:
      sb.apply1(x, i)
    for i in reg:
      sb.apply1(h, i)
  return sb


def grover_initial_index(nbits):
  """Basis index of |0..0>|1>|0..0> on 2*nbits qubits (ancilla = qubit nbits)."""
  n = 2 * nbits
  return 1 << (n - 1 - nbits)


def grover_recurrence(nbits, iterations):
  """The four distinct amplitudes after `iterations` (SURVEY 8c analytic oracle).

  Returns (c_m0, c_m1, c_u0, c_u1): marked/unmarked x ancilla 0/1.
  """
  big_n = 2 ** nbits
  cm0 = cu0 = 1 / math.sqrt(2 * big_n)
  cm1 = cu1 = -1 / math.sqrt(2 * big_n)
  for _ in range(iterations):
    cm0, cm1 = cm1, cm0
    mu = (cm1 + (big_n - 1) * cu1) / big_n
    cm1 -= 2 * mu
    cu1 -= 2 * mu
  return cm0, cm1, cu0, cu1


def qft_analytic(nbits, x, idx):
  """QFT of basis state x at indices idx: exp(2 pi i bitrev(x) k / N)/sqrt(N)."""
  xr = int(format(x, f'0{nbits}b')[::-1], 2)
  idx = np.asarray(idx, dtype=np.uint64)
  n = 1 << nbits
  # (xr * k) mod N in exact integer arithmetic (python ints for safety)
  ph = np.array([(xr * int(k)) % n for k in idx.ravel()], dtype=np.float64) / n
  return (np.exp(2j * np.pi * ph) / math.sqrt(n)).reshape(idx.shape)
