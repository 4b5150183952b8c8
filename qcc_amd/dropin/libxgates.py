"""Drop-in ``libxgates`` module: put this directory on PYTHONPATH and the
reference's unmodified src/lib/circuit.py binds it at import
(``import libxgates as xgates; apply1 = xgates.apply1; applyc = xgates.applyc``,
circuit.py:36-41) -- every gate of every reference algorithm then runs on the
MI355X through the C-ABI (qh_host_apply1 / qh_host_applyc).

Signatures are the reference's (src/lib/xgates.cc:89-107,126-145), positional:
    apply1(psi, gate, nbits, tgt, bit_width) -> None
    applyc(psi, gate, nbits, ctl, tgt, bit_width) -> None
psi is updated in place.  This literal form moves the state over PCIe per gate;
qcc_amd.lib.circuit keeps it resident instead.
"""
from qcc_amd.lib import backend as _backend


def apply1(psi, gate, nbits, tgt, bit_width):
    _backend.host_executor().apply1(psi, gate, nbits, tgt, bit_width)


def applyc(psi, gate, nbits, ctl, tgt, bit_width):
    _backend.host_executor().applyc(psi, gate, nbits, ctl, tgt, bit_width)
