"""Small numeric helpers used around the hot path: bit strings <-> integers,
binary fractions, Bloch-sphere coordinates, pretty printing of angles.

Same function names and results as the reference's ``src/lib/helper.py``; the
bit conventions are the ones the native boundary uses (most significant bit
first: qubit 0 is the top index bit, ``src/lib/xgates.cc:26``).
"""
import itertools
import math

import numpy as np

_PI_MULTIPLES = (1, 2, 3)
_PI_DENOMS = [d for d in range(-127, 128) if d]


def val2bits(val, nbits):
    """Integer -> list of bits, MSB first, at least `nbits` long.

    Mirrors ``format(val, '0{nbits}b')`` of the reference: a value that does not
    fit is returned with MORE bits, never truncated (minimum_finding.py relies on it)."""
    val = int(val)
    width = max(int(nbits), val.bit_length())
    return [(val >> shift) & 1 for shift in range(width - 1, -1, -1)]


def bits2val(bits):
    """List of bits (MSB first) -> integer."""
    out = 0
    for bit in bits:
        out = 2 * out + int(bit)
    return out


def bitprod(nbits):
    """Iterate over all 2^nbits bit tuples in counting order."""
    return itertools.product((0, 1), repeat=nbits)


def frac2bits(val, nbits):
    """The first `nbits` binary digits of val in [0, 1)."""
    assert val < 1.0, 'frac2bits: value must be strictly < 1.0'
    digits = []
    rest = val
    while len(digits) < nbits:
        rest *= 2
        digits.append(int(rest))
        rest -= digits[-1]
    return digits


def bits2frac(bits):
    """Value of the binary fraction 0.b0b1b2..."""
    return sum(b / float(1 << (pos + 1)) for pos, b in enumerate(bits))


def pi_fractions(val, pi='pi'):
    """'pi/4', '-3*pi/8', ... when val is such a multiple of pi, else str(val)."""
    if val is None:
        return ''
    if val == 0:
        return '0'
    for mult in _PI_MULTIPLES:
        for denom in _PI_DENOMS:
            if not math.isclose(val, mult * math.pi / denom):
                continue
            text = pi if mult == 1 else f'{mult}*{pi}'
            if abs(denom) != 1:
                text += f'/{abs(denom)}'
            return text if denom > 0 else '-' + text
    return f'{val}'


# -- Bloch sphere -------------------------------------------------------------------------
def density_to_cartesian(rho):
    """(x, y, z) of the Bloch vector of a single-qubit density matrix."""
    off_diag, top = rho[1, 0], rho[0, 0]
    coords = (2.0 * off_diag.real, 2.0 * off_diag.imag, 2.0 * top - 1.0)
    return tuple(np.real(c) for c in coords)


def qubit_to_bloch(psi):
    sync = getattr(psi, '_sync_host', None)      # (a State with gates on its device mirror: qcc_amd/lib/state.py)
    if sync is not None:
        sync()
    psi = np.asarray(psi)
    return density_to_cartesian(np.outer(psi, psi.conj()))


def dump_bloch(x, y, z):
    print('x: {:.2f}, y: {:.2f}, z: {:.2f}'.format(x, y, z))


def qubit_dump_bloch(psi):
    dump_bloch(*qubit_to_bloch(psi))
