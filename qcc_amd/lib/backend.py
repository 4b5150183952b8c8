"""Where qcc_amd.lib sends gate applications.

Two seams, both defaulting to the HIP engine (no CPU fallback lives here):

* host executor -- applies one gate to a host NumPy buffer in place; the
  equivalent of the reference's ``libxgates.apply1/applyc`` module functions
  (src/lib/circuit.py:36-41).  Default: the C-ABI drop-in qh_host_apply1/applyc.
* device factory -- creates the HBM-resident state a ``circuit.qc`` works on.
  Default: qcc_amd.device.DeviceState with fused sweeps.

Tests may install substitutes (e.g. a recording or oracle-backed executor) with
set_host_executor / set_device_factory; product code never does.
"""
import ctypes
import os

import numpy as np

from qcc_amd import gates as _gates
from qcc_amd import native

_dp = ctypes.POINTER(ctypes.c_double)


class HipHostExecutor:
    """libxgates-compatible callables over the C-ABI (H2D, kernel, D2H)."""

    def __init__(self):
        self.lib = native.load()

    @staticmethod
    def _check_buffer(psi, bit_width):
        want = np.complex128 if bit_width == 128 else np.complex64
        if not isinstance(psi, np.ndarray) or psi.dtype != want or not psi.flags.c_contiguous:
            # the reference silently updates a converted temporary and drops it
            # (xgates.cc:75-77); that is a bug magnet, so say so.
            raise TypeError(f'psi must be a C-contiguous {np.dtype(want).name} array for bit_width={bit_width}')

    def apply1(self, psi, gate, nbits, tgt, bit_width=128):
        self._check_buffer(psi, bit_width)
        g = _gates.as8(gate)
        native.check(self.lib.qh_host_apply1(psi.ctypes.data, g.ctypes.data_as(_dp), int(nbits), int(tgt),
                                             int(bit_width)))

    def applyc(self, psi, gate, nbits, ctl, tgt, bit_width=128):
        self._check_buffer(psi, bit_width)
        g = _gates.as8(gate)
        native.check(self.lib.qh_host_applyc(psi.ctypes.data, g.ctypes.data_as(_dp), int(nbits), int(ctl),
                                             int(tgt), int(bit_width)))


_host_executor = None
_device_factory = None


def host_executor():
    global _host_executor
    if _host_executor is None:
        _host_executor = HipHostExecutor()
    return _host_executor


def set_host_executor(ex):
    global _host_executor
    _host_executor = ex


def _default_device_factory(nbits, bit_width):
    fusion = native.QH_FUSE_OFF if os.environ.get('QCC_FUSION', '1') == '0' else native.QH_FUSE_SWEEP
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1 and nbits >= 2 * max(1, world.bit_length() - 1) + 2:
        # one process per GPU (torchrun): the register shards by its top log2(WORLD_SIZE) index
        # bits and every rank runs the same script (north_star: "the circuit.qc() Python API so
        # every algorithm runs unmodified"); registers too small to shard stay replicated
        from qcc_amd import sharded
        return sharded.ShardedDevice(nbits, bit_width, fusion=fusion)
    from qcc_amd import device
    return device.DeviceState(nbits, bit_width, fusion=fusion,
                              device=int(os.environ.get('LOCAL_RANK', '0')) if world > 1 else 0)


def make_host_mapped_state(nbits, bit_width):
    """The state in pinned host memory the GPU works on in place (circuit.qc(alias_psi=True))."""
    if _host_mapped_factory is not None:
        return _host_mapped_factory(nbits, bit_width)
    from qcc_amd import device
    return device.DeviceState(nbits, bit_width, fusion=native.QH_FUSE_SWEEP, host_mapped=True)


_host_mapped_factory = None


def set_host_mapped_factory(factory):
    global _host_mapped_factory
    _host_mapped_factory = factory


def make_device_state(nbits, bit_width):
    return (_device_factory or _default_device_factory)(nbits, bit_width)


def set_device_factory(factory):
    global _device_factory
    _device_factory = factory
