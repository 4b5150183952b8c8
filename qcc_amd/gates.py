"""2x2 gate matrices fed to the hot path (complex128, row-major).

The native boundary only ever sees four matrix entries, so parity
with the reference starts with producing the same entries.  Each constructor
evaluates the same closed form as the reference constructor it cites
(/root/reference/src/lib/ops.py), so the doubles are identical; exact zeros are
kept exact because the engine classifies gates by them.
"""
import cmath
import math

import numpy as np

_SQRT1_2 = 1 / np.sqrt(2)


def _m(a, b, c, d):
  return np.array([[a, b], [c, d]], dtype=np.complex128)


def identity():
  return _m(1.0, 0.0, 0.0, 1.0)


def pauli_x():  # ops.py:114-115
  return _m(0.0, 1.0, 1.0, 0.0)


def pauli_y():  # ops.py:118-119
  return _m(0.0, -1.0j, 1.0j, 0.0)


def pauli_z():  # ops.py:122-123
  return _m(1.0, 0.0, 0.0, -1.0)


def hadamard():  # ops.py:130-132
  return (_SQRT1_2 * np.array([[1.0, 1.0], [1.0, -1.0]])).astype(np.complex128)


def sgate():  # ops.py:136-142
  return _m(1.0, 0.0, 0.0, 1.0j)


def tgate():  # ops.py:146-147
  return _m(1.0, 0.0, 0.0, cmath.exp(cmath.pi * 1j / 4))


def vgate():  # ops.py:152-154  sqrt(X)
  return (0.5 * np.array([(1 + 1j, 1 - 1j), (1 - 1j, 1 + 1j)])).astype(np.complex128)


def yroot():  # ops.py:158-162  sqrt(Y)
  return (0.5 * np.array([(1 + 1j, -1 - 1j), (1 + 1j, 1 + 1j)])).astype(np.complex128)


def u1(lam):  # ops.py:166-167
  return _m(1.0, 0.0, 0.0, cmath.exp(1j * lam))


def rk(k):  # ops.py:182-183
  return u1(2 * math.pi / (2 ** k))


def u3(theta, phi, lam):  # ops.py:171-178
  return _m(np.cos(theta / 2), -cmath.exp(1j * lam) * np.sin(theta / 2),
            cmath.exp(1j * phi) * np.sin(theta / 2),
            cmath.exp(1j * (phi + lam)) * np.cos(theta / 2))


def rotation(v, theta):  # ops.py:190-199
  v = np.asarray(v, dtype=float)
  if v.shape != (3,) or not math.isclose(v @ v, 1) or not np.all(np.isreal(v)):
    raise ValueError('Rotation vector v must be a 3D real unit vector.')
  return (np.cos(theta / 2) * identity() - 1j * np.sin(theta / 2) *
          (v[0] * pauli_x() + v[1] * pauli_y() + v[2] * pauli_z()))


def rx(theta):
  return rotation([1.0, 0.0, 0.0], theta)


def ry(theta):
  return rotation([0.0, 1.0, 0.0], theta)


def rz(theta):
  return rotation([0.0, 0.0, 1.0], theta)


def adjoint(g):
  return np.conj(np.asarray(g).T).astype(np.complex128)


def as8(gate):
  """Any 2x2 / 4-element complex gate -> contiguous float64[8] for the C-ABI."""
  g = np.ascontiguousarray(np.asarray(gate, dtype=np.complex128).reshape(4))
  return g.view(np.float64)
