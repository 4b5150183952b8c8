"""Exact product-state tracker: an analytic oracle for circuits that never entangle.

The QFT of a BASIS state keeps the register a product of single-qubit states all the way
through (src/lib/circuit.py:320-328: H(i), then CU1(i -> j) for j < i -- the target j has not
seen its H yet, so it is still |0> or |1> and the controlled phase is a plain phase on the
control's |1> component).  Tracking the n two-vectors costs O(gates) and gives any amplitude
of any intermediate state as a product of n numbers: the oracle for sizes no reference can run
("no reference above 30 qubits"), in particular for the pieces of the 36-qubit
sharded QFT (config 5) that one GPU holds.  Checked against the C oracle at small n
(tests/test_product_oracle_cpu.py).  Test infrastructure only.
"""
import numpy as np

NO_CTL = -(2 ** 31)


class ProductState:
  def __init__(self, nbits, basis_index):
    self.n = int(nbits)
    self.v = np.zeros((self.n, 2), dtype=np.complex128)   # v[q] = state of reference qubit q
    for q in range(self.n):
      self.v[q, (int(basis_index) >> (self.n - 1 - q)) & 1] = 1.0

  def _basis_value(self, q):
    a, b = self.v[q]
    if b == 0 and abs(abs(a) - 1) < 1e-12:
      return 0
    if a == 0 and abs(abs(b) - 1) < 1e-12:
      return 1
    return None

  def apply1(self, g4, q):
    g = np.asarray(g4, dtype=np.complex128).reshape(2, 2)
    self.v[q] = g @ self.v[q]

  def applyc(self, g4, c, t):
    g = np.asarray(g4, dtype=np.complex128).reshape(2, 2)
    bc = self._basis_value(c)
    if bc is not None:                       # control is a basis state: plain (un)conditional gate
      if bc == 1:
        self.v[t] = g @ self.v[t]
      return
    bt = self._basis_value(t)
    if bt is not None and g[0, 1] == 0 and g[1, 0] == 0:
      self.v[c, 1] *= g[bt, bt]              # diagonal gate on a basis-state target: a phase on the control's |1>
      return
    raise ValueError(f'gate ({c}->{t}) would entangle: not a product circuit')

  def run(self, ops, g8, first=0, last=None):
    gc = np.ascontiguousarray(g8, dtype=np.float64).view(np.complex128).reshape(-1, 4)
    last = len(ops) if last is None else last
    for k in range(first, last):
      c, t = int(ops[k, 0]), int(ops[k, 1])
      if c == NO_CTL:
        self.apply1(gc[k], t)
      else:
        self.applyc(gc[k], c, t)

  def amplitudes(self, idx):
    """Amplitudes at global LOGICAL indices idx (qubit q = index bit n-1-q)."""
    idx = np.asarray(idx, dtype=np.uint64)
    out = np.ones(idx.shape, dtype=np.complex128)
    for q in range(self.n):
      bit = ((idx >> np.uint64(self.n - 1 - q)) & np.uint64(1)).astype(np.int64)
      out *= self.v[q][bit]
    return out

  def factors(self):
    """[(1, two-vector)] per qubit, most significant first: input of qh_init_product."""
    return [(1, self.v[q].copy()) for q in range(self.n)]
