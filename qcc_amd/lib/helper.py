"""Bit/fraction helpers (API mirror of /root/reference/src/lib/helper.py)."""
import itertools
import math

import numpy as np


def bitprod(nbits):
    """All bit tuples of length nbits, big-endian counting order."""
    yield from itertools.product((0, 1), repeat=nbits)


def bits2val(bits):
    """[1, 1, 0] -> 6 (most significant bit first)."""
    val = 0
    for b in bits:
        val = (val << 1) | int(b)
    return val


def val2bits(val, nbits):
    """6, 3 -> [1, 1, 0].  Like the reference's format(val, '0{nbits}b') a value
    that needs more than nbits bits is NOT truncated (helper.py:26-31)."""
    width = max(int(nbits), int(val).bit_length())
    return [(int(val) >> (width - 1 - i)) & 1 for i in range(width)]


def bits2frac(bits):
    """Binary fraction 0.b0 b1 b2 ..."""
    return sum(bit * 2.0 ** (-i - 1) for i, bit in enumerate(bits))


def frac2bits(val, nbits):
    """First nbits binary digits of a fraction in [0, 1)."""
    assert val < 1.0, 'frac2bits: value must be strictly < 1.0'
    out = []
    for _ in range(nbits):
        val *= 2
        digit = int(val)
        out.append(digit)
        val -= digit
    return out


def density_to_cartesian(rho):
    """Bloch-sphere (x, y, z) of a 2x2 density matrix."""
    a, b = rho[0, 0], rho[1, 0]
    return np.real(2.0 * b.real), np.real(2.0 * b.imag), np.real(2.0 * a - 1.0)


def qubit_to_bloch(psi):
    return density_to_cartesian(np.outer(psi, np.conj(psi)))


def dump_bloch(x, y, z):
    print(f'x: {x:.2f}, y: {y:.2f}, z: {z:.2f}')


def qubit_dump_bloch(psi):
    dump_bloch(*qubit_to_bloch(psi))


def pi_fractions(val, pi='pi'):
    """Render val as a small multiple/fraction of pi when it is one."""
    if val is None:
        return ''
    if val == 0:
        return '0'
    for mult in range(1, 4):
        for denom in range(-128, 128):
            if denom and math.isclose(val, mult * math.pi / denom):
                head = '' if mult == 1 else f'{mult}*'
                sign = '-' if denom < 0 else ''
                tail = '' if abs(denom) == 1 else f'/{abs(denom)}'
                return f'{sign}{head}{pi}{tail}'
    return f'{val}'
