"""State vectors and registers (API mirror of /root/reference/src/lib/state.py).

``State`` is a host-side ndarray (so ``psi[i]``, ``np.abs(psi)``, slicing keep
working exactly as in the reference).  Its two hot methods, ``apply1`` and
``applyc`` (state.py:80-125, the pure-Python specification of the native loops),
run on the GPU through the C-ABI drop-in instead of Python loops.
"""
import cmath
import math
import random

import numpy as np

from qcc_amd.lib import backend
from qcc_amd.lib import helper
from qcc_amd.lib import tensor


class State(tensor.Tensor):
    """Amplitudes of an n-qubit pure state, qubit 0 = most significant index bit."""

    def __array_finalize__(self, obj):
        super().__array_finalize__(obj)
        self.basis_index = None  # set by the basis-state constructors below

    def density(self):
        return tensor.Tensor(np.outer(self, self.conj()))

    def adjoint(self):
        return self.conj().transpose()

    def normalize(self):
        norm2 = np.conj(self) @ self
        assert not np.allclose(norm2, 0.0, atol=1e-6), 'Normalizing to 0-probability state'
        return self / np.sqrt(np.real(norm2))

    def ampl(self, *bits):
        return self[helper.bits2val(bits)]

    def prob(self, *bits):
        a = self.ampl(*bits)
        return np.real(a.conj() * a)

    def phase(self, *bits):
        return math.degrees(cmath.phase(self.ampl(*bits)))

    def diff(self, psi, dump=True):
        """Element-wise comparison (abs tol 1e-4); prints the differences."""
        same = True
        for idx, val in enumerate(self):
            if cmath.isclose(val, psi[idx], abs_tol=1e-4):
                continue
            same = False
            if dump:
                print(f'State{helper.val2bits(idx, self.nbits)} (|{idx}>):{val:+.3f}  {psi[idx]:+.3f}')
        return same

    def maxprob(self):
        """(bits, probability) of the most likely basis state (state.py:60-78)."""
        idx = int(np.abs(self).argmax())
        p = np.real(self[idx].conj() * self[idx])
        return helper.val2bits(idx, self.nbits), p

    # -- the hot path: same call surface as state.py:80,102 -------------------------
    def _exec_buffer(self):
        if not (self.flags.c_contiguous and self.flags.writeable):
            raise ValueError('State.apply1/applyc need a contiguous, writeable state')
        return self.view(np.ndarray)

    def apply1(self, gate, index):
        n = self.nbits
        if not 0 <= index < n:
            raise ValueError(f'apply1: qubit {index} out of range for {n} qubits')
        width = 128 if self.dtype == np.complex128 else 64
        backend.host_executor().apply1(self._exec_buffer(), np.asarray(gate).reshape(4), n, index, width)

    def applyc(self, gate, control, target):
        n = self.nbits
        if not 0 <= target < n:
            raise ValueError(f'applyc: qubit {target} out of range for {n} qubits')
        width = 128 if self.dtype == np.complex128 else 64
        backend.host_executor().applyc(self._exec_buffer(), np.asarray(gate).reshape(4), n, control, target,
                                       width)

    def dump(self, desc=None, prob_only=True):
        """Print the basis states with non-negligible probability."""
        n = self.nbits
        digits = int(math.log10(2 ** n)) + 1
        if desc:
            print('|' + ''.join(str(i % 10) for i in range(n)) + f"> '{desc}'")
        rows = []
        for bits in helper.bitprod(n):
            p = self.prob(*bits)
            if prob_only and p < 10e-6:
                continue
            s = ''.join(str(b) for b in bits)
            rows.append(f'|{s}> (|{int(s, 2):{digits}d}>):  ampl: {self.ampl(*bits):+.2f} '
                        f'prob: {p:.2f} Phase: {self.phase(*bits):5.1f}')
        rows.sort()
        print(*rows, sep='\n')


# -- constructors ----------------------------------------------------------------------
def qubit(alpha=None, beta=None):
    """Single-qubit state alpha|0> + beta|1>; the missing one is derived."""
    if alpha is None and beta is None:
        raise ValueError('alpha, beta, or both, need to be specified')
    if beta is None:
        beta = np.sqrt(1.0 - np.real(np.conj(alpha) * alpha))
    if alpha is None:
        alpha = np.sqrt(1.0 - np.real(np.conj(beta) * beta))
    total = np.real(np.conj(alpha) * alpha) + np.real(np.conj(beta) * beta)
    assert math.isclose(total, 1.0), 'Qubit probabilities not equal to 1.'
    return State([alpha, beta])


def _basis(nbits, index):
    vec = np.zeros(1 << nbits, dtype=tensor.tensor_type())
    vec[index] = 1
    out = State(vec)
    out.basis_index = index
    return out


def zeros_or_ones(d=1, idx=0):
    assert d > 0, 'Need to specify at least 1 qubit'
    return _basis(d, idx)


def zeros(d=1):
    return zeros_or_ones(d, 0)


def ones(d=1):
    return zeros_or_ones(d, 2 ** d - 1)


def _product(a, b, d):
    return State([a, b]).kpow(d)


def plus(d=1):
    return _product(1 / np.sqrt(2), 1 / np.sqrt(2), d)


def minus(d=1):
    return _product(1 / np.sqrt(2), -1 / np.sqrt(2), d)


def plusi(d=1):
    return _product(1 / np.sqrt(2), 1j / np.sqrt(2), d)


def minusi(d=1):
    return _product(1 / np.sqrt(2), -1j / np.sqrt(2), d)


def bitstring(*bits):
    arr = np.asarray(bits)
    assert len(arr), 'Need to specify at least 1 qubit'
    assert ((arr == 1) | (arr == 0)).all(), 'Bits must be 0 or 1'
    return _basis(len(bits), helper.bits2val(bits))


def rand_bits(n):
    return bitstring(*[random.randint(0, 1) for _ in range(n)])


class Reg(list):
    """A named run of consecutive qubit indices plus its initial bit values."""

    def __init__(self, size, init=None, global_reg=0):
        super().__init__(range(global_reg, global_reg + size))
        self.val = [0] * size
        if init:
            if isinstance(init, int):
                init = format(init, f'0{size}b')
            if isinstance(init, (str, tuple, list)):
                for pos, v in enumerate(init):
                    if v in ('1', 1):
                        self.val[pos] = 1

    def __str__(self):
        return '|' + ''.join(str(v) for v in self.val) + '>'

    def psi(self):
        return bitstring(*self.val)
