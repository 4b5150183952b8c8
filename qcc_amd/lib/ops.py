"""Operators and gate constructors (API mirror of /root/reference/src/lib/ops.py).

The 2x2 constructors delegate to qcc_amd.gates, which evaluates the same closed
forms as ops.py:110-207, so the matrix entries handed to the hot path are the
reference's doubles.  The dense multi-qubit builders (ControlledU, Cnot, Swap,
...) and Operator application by full matrices are kept for small didactic
sizes only -- O(4^n), host NumPy, not part of the accelerated path.
"""
import math

import numpy as np

from qcc_amd import gates as _g
from qcc_amd.lib import helper
from qcc_amd.lib import state
from qcc_amd.lib import tensor


class Operator(tensor.Tensor):
    """A square matrix acting on qubits; calling it applies it."""

    def __new__(cls, input_array, name=None):
        return super().__new__(cls, input_array, name)

    def adjoint(self):
        return self.__class__(np.conj(self.transpose()))

    def dump(self, desc=None, digits=3):
        np.set_printoptions(precision=digits)
        if desc:
            print(f'{desc} ({self.nbits}-qubit(s) operator)')
        print(self)

    def apply(self, arg, idx):
        """op(state, idx): pad with identities and multiply.  op(other_op, idx):
        compose, with the reference's convention x(y) == y @ x (ops.py:59-104)."""
        if isinstance(arg, Operator):
            other_bits = arg.nbits
            if idx > 0:
                arg = Identity().kpow(idx) * arg
            if self.nbits > arg.nbits:
                arg = arg * Identity().kpow(self.nbits - idx - other_bits)
            assert self.nbits == arg.nbits, 'Misatched dimensions.'
            return arg @ self
        assert isinstance(arg, state.State), 'Error, expected State.'
        full = self
        if idx > 0:
            full = Identity().kpow(idx) * full
        trailing = arg.nbits - idx - self.nbits
        if trailing > 0:
            full = full * Identity().kpow(trailing)
        return state.State(np.matmul(full, arg))

    def __call__(self, arg, idx=0):
        return self.apply(arg, idx)


def _gate(matrix, name, d):
    return Operator(matrix, name).kpow(d)


def Identity(d=1):
    return _gate(_g.identity(), 'Id', d)


def PauliX(d=1):
    return _gate(_g.pauli_x(), 'X', d)


def PauliY(d=1):
    return _gate(_g.pauli_y(), 'Y', d)


def PauliZ(d=1):
    return _gate(_g.pauli_z(), 'Z', d)


def Pauli(d=1):
    return Identity(d), PauliX(d), PauliY(d), PauliZ(d)


def Hadamard(d=1):
    return _gate(_g.hadamard(), 'H', d)


def Phase(d=1):
    return _gate(_g.sgate(), 'S', d)


def Sgate(d=1):
    return Phase(d)


def Tgate(d=1):
    return _gate(_g.tgate(), None, d)


def Vgate(d=1):
    return _gate(_g.vgate(), 'V', d)


def Yroot(d=1):
    return _gate(_g.yroot(), 'YRoot', d)


def U1(lam, d=1):
    return _gate(_g.u1(lam), 'U1', d)


def U3(theta, phi, lam, d=1):
    return _gate(_g.u3(theta, phi, lam), 'U3', d)


def Rk(k, d=1):
    return U1(2 * math.pi / (2 ** k)).kpow(d)


def Rotation(vparm, theta, name):
    return Operator(_g.rotation(vparm, theta), name + f'({theta:.3f})')


def RotationX(theta):
    return Rotation([1.0, 0.0, 0.0], theta, 'Rx')


def RotationY(theta):
    return Rotation([0.0, 1.0, 0.0], theta, 'Ry')


def RotationZ(theta):
    return Rotation([0.0, 0.0, 1.0], theta, 'Rz')


def _projector(nbits, which, name):
    dim = 2 ** nbits
    m = np.zeros((dim, dim))
    m[which, which] = 1
    return Operator(m, name)


def ZeroProjector(nbits):
    return _projector(nbits, 0, 'P0')


def OneProjector(nbits):
    return _projector(nbits, 2 ** nbits - 1, 'P1')


def ControlledU(idx0, idx1, u):
    """Full matrix of U on idx1 controlled by idx0 (only their distance and order matter)."""
    assert idx0 != idx1, 'Control / controlled must not be equal.'
    gap = Identity(abs(idx1 - idx0) - 1)
    idle = Identity().kpow(u.nbits)
    p0, p1 = ZeroProjector(1), OneProjector(1)
    if idx1 > idx0:
        return p0 * gap * idle + p1 * gap * u
    return idle * gap * p0 + u * gap * p1


def Cnot(idx0=0, idx1=1):
    return ControlledU(idx0, idx1, PauliX())


def Cnot0(idx0=0, idx1=1):
    """Cnot controlled by |0>: X on the control before and after."""
    if idx1 > idx0:
        flip = PauliX() * Identity(idx1 - idx0)
    else:
        flip = Identity(idx0 - idx1) * PauliX()
    return flip @ ControlledU(idx0, idx1, PauliX()) @ flip


def Swap(idx0=0, idx1=1):
    # pylint: disable=arguments-out-of-order
    return Cnot(idx1, idx0) @ Cnot(idx0, idx1) @ Cnot(idx1, idx0)


def Toffoli(idx0, idx1, idx2):
    return ControlledU(idx0, idx1, Cnot(idx1, idx2))


def Measure(psi, idx, tostate=0, collapse=True):
    """P(qubit idx == tostate) and, optionally, the collapsed normalised state.

    Same results as the reference's projector-on-density-matrix formulation
    (ops.py:426-460) but O(2^n): a masked norm instead of a 4^n density matrix."""
    n = psi.nbits
    bit = n - 1 - idx
    amp = np.asarray(tensor.host_current(psi))
    keep = ((np.arange(amp.shape[0]) >> bit) & 1) == (1 if tostate else 0)
    prob = float(np.real(np.vdot(amp[keep], amp[keep])))
    if not collapse:
        return prob, psi
    kept = np.where(keep, amp, 0)
    norm = float(np.linalg.norm(kept))
    assert norm > 1e-10, 'Measurement collapses to 0.0.'
    return prob, state.State(kept / norm)


# -- full-matrix helpers kept for the didactic small-n code paths (host only) --------
def OracleUf(nbits, f):
    """Permutation matrix |x>|y> -> |x>|y xor f(x)> for an (nbits-1)-bit x."""
    dim = 2 ** nbits
    u = np.zeros((dim, dim))
    for row in range(dim):
        bits = helper.val2bits(row, nbits)
        col = helper.bits2val(bits[:-1] + [bits[-1] ^ int(f(bits[:-1]))])
        u[row, col] = 1.0
    op = Operator(u)
    assert op.is_unitary(), 'Constructed non-unitary operator.'
    return op


def Qft(nbits, swap=True):
    """QFT as one matrix: per qubit a Hadamard then controlled R_k's, optional reversal."""
    op = Identity(nbits)
    for q in range(nbits):
        op = op(Hadamard(), q)
        for k in range(2, nbits - q + 1):
            op = op(ControlledU(q + k - 1, q, Rk(k)), q)
    if swap:
        for q in range(nbits // 2):
            op = op(Swap(q, nbits - q - 1), q)
    assert op.is_unitary(), 'Constructed non-unitary operator.'
    return op


def PhaseEstimation(op, psi, nbits_phase, target, offset=0):
    """Controlled powers op^(2^k) from the phase register onto `target`."""
    power = op
    for q in reversed(range(nbits_phase)):
        psi = ControlledU(q + offset, target, power)(psi, q + offset)
        power = power(power)
    return psi


def TraceOutSingle(rho, index):
    """Partial trace over one qubit of a density matrix."""
    nbits = int(math.log2(rho.shape[0]))
    assert 0 <= index < nbits, 'TraceOutSingle: Invalid index.'
    left, right = 2 ** index, 2 ** (nbits - index - 1)
    t = np.asarray(rho).reshape(left, 2, right, left, 2, right)
    reduced = np.einsum('aibcid->abcd', t).reshape(left * right, left * right)
    return Operator(reduced)


def TraceOut(rho, index_set):
    """Partial trace over several qubits (indices refer to the original numbering)."""
    for pos, q in enumerate(index_set):
        rho = TraceOutSingle(rho, q)
        for later in range(pos + 1, len(index_set)):
            index_set[later] -= 1
    return rho
