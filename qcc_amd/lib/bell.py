"""Bell / GHZ / W states (API mirror of /root/reference/src/lib/bell.py), small sizes."""
import numpy as np

from qcc_amd.lib import ops
from qcc_amd.lib import state


def bell_state(a, b):
    """One of the four Bell states |b_ab>: H on qubit 0, then CNOT(0,1) on |ab>."""
    if a not in (0, 1) or b not in (0, 1):
        raise ValueError('Bell state arguments are bits and must be 0 or 1.')
    psi = state.bitstring(a, b)
    psi = ops.Hadamard()(psi)
    return ops.Cnot()(psi)


def ghz_state(nbits):
    """(|0..0> + |1..1>)/sqrt(2)."""
    psi = np.zeros(2 ** nbits)
    psi[0] = psi[-1] = 1 / np.sqrt(2)
    return state.State(psi)


def w_state():
    """(|001> + |010> + |100>)/sqrt(3)."""
    psi = np.zeros(8)
    psi[[1, 2, 4]] = 1 / np.sqrt(3)
    return state.State(psi)
