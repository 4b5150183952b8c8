"""Tensor: the ndarray subclass every state/operator derives from.

API mirror of /root/reference/src/lib/tensor.py (class Tensor :49-138, dtype
policy :28-46).  The precision policy is kept: 64 -> complex64 (the reference's
default), 128 -> complex128; select with ``--tensor_width`` when absl is in use,
with :func:`set_tensor_width`, or with the environment variable
``QCC_TENSOR_WIDTH``.
"""
import math
import os

import numpy as np

_width_override = None

try:  # absl is optional: the reference algorithms use it, this package does not need it
    from absl import flags as _flags
    try:
        _flags.DEFINE_integer('tensor_width', 64, 'Bitwidth of FP numbers (64 or 128)')
    except Exception:  # pylint: disable=broad-except  (already defined by someone else)
        pass
except Exception:  # pylint: disable=broad-except
    _flags = None


def set_tensor_width(width):
    """Force the complex width (64 or 128); None returns to flag/env/default."""
    global _width_override
    assert width in (None, 64, 128)
    _width_override = width


def tensor_width():
    """Bit width of one complex amplitude: 64 or 128 (tensor.py:31-37)."""
    if _width_override is not None:
        return _width_override
    if _flags is not None:
        try:
            return int(_flags.FLAGS.tensor_width)
        except Exception:  # pylint: disable=broad-except  (flags not parsed: REPL use)
            pass
    env = os.environ.get('QCC_TENSOR_WIDTH')
    return int(env) if env else 64


def tensor_type():
    """NumPy dtype for the current width (tensor.py:42-46)."""
    width = tensor_width()
    assert width in (64, 128), 'tensor_width must be 64 or 128'
    return np.complex128 if width == 128 else np.complex64


def host_current(x):
    """x, with its host memory current: a State whose gates still sit on its device mirror (qcc_amd/lib/state.py) brings
    them home.  np.asarray / np.array hand an ndarray subclass's memory out without asking it, so the library calls this
    wherever it takes an array apart in that way."""
    sync = getattr(x, '_sync_host', None)
    if sync is not None:
        sync()
    return x


class Tensor(np.ndarray):
    """A NumPy array with quantum-flavoured helpers; ``*`` is the Kronecker product."""

    def __new__(cls, input_array, op_name=None):
        obj = np.asarray(host_current(input_array), dtype=tensor_type()).view(cls)
        obj.name = op_name
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.name = getattr(obj, 'name', None)

    @property
    def nbits(self):
        return int(math.log2(self.shape[0]))

    # -- predicates ---------------------------------------------------------------
    def is_close(self, arg, tolerance=1e-6):
        return bool(np.allclose(self, arg, atol=tolerance))

    def is_hermitian(self):
        if self.ndim != 2 or self.shape[0] != self.shape[1]:
            return False
        return self.is_close(np.conj(self.transpose()))

    def is_unitary(self):
        prod = Tensor(np.conj(self.transpose()) @ self)
        return prod.is_close(Tensor(np.eye(self.shape[0])))

    def is_density(self):
        return self.is_hermitian() and not np.trace(self) > 1.0

    def is_pure(self):
        if not self.is_density():
            raise ValueError('ispure() can only be applied to a density matrix.')
        return bool(np.allclose(np.real(np.trace(self @ self)), 1.0))

    def is_permutation(self):
        arr = np.asarray(self)
        return bool(arr.ndim == 2 and arr.shape[0] == arr.shape[1]
                    and (arr.sum(axis=0) == 1).all() and (arr.sum(axis=1) == 1).all()
                    and ((arr == 1) | (arr == 0)).all())

    # -- tensor products ------------------------------------------------------------
    def kron(self, arg):
        left = self.name or '?'
        right = getattr(arg, 'name', None) or '?'
        return self.__class__(np.kron(self, arg), left + '*' + right)

    def __mul__(self, arg):
        return self.kron(arg)

    def kpow(self, n):
        """n-fold Kronecker power; kpow(0) is the scalar 1."""
        if n == 0:
            return self.__class__(1.0)
        acc = np.asarray(self)
        for _ in range(n - 1):
            acc = np.kron(acc, self)
        label = self.name if n == 1 else (self.name or '?') + f'^{n}'
        return self.__class__(acc, label)
