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


# ---- device-state pool ------------------------------------------------------------------------------------------
# A script that builds one circuit after another (the reference's algorithms do: src/lib/circuit.py:71-101 creates a
# qc per experiment) paid for two fresh 16-GiB buffers per 30-qubit circuit: allocation + first touch ~ 3.8 s cold,
# 12-17 ms warm (`single_shot_ms`).  Released states of the default factory are kept here, keyed by
# (qubits, width), and handed to the next circuit of the same shape -- with their second buffer, op buffers and cached
# plans.  Bounded: at most QCC_POOL_STATES handles (default 2) of at most QCC_POOL_MAX_QUBITS qubits (default 31: a
# 34-qubit state is never parked); drop_device_pool() frees everything (also done when an allocation fails).
_pool = {}
_pool_order = []


def _pool_limits():
    return int(os.environ.get('QCC_POOL_STATES', '2')), int(os.environ.get('QCC_POOL_MAX_QUBITS', '31'))


def drop_device_pool():
    for dev in _pool_order:
        try:
            dev._parked = False
            dev.close()
        except Exception:  # pylint: disable=broad-except
            pass
    _pool.clear()
    del _pool_order[:]


def release_device_state(dev):
    """Called by circuit.qc when it is done with a device state: park it for the next circuit, or close it."""
    keep, max_bits = _pool_limits()
    poolable = (_device_factory is None and keep > 0 and type(dev).__name__ == 'DeviceState' and dev.nbits <= max_bits
                and getattr(dev, 'h', None) and int(os.environ.get('WORLD_SIZE', '1')) == 1)
    if not poolable:
        dev.close()
        return
    try:
        native.check(dev.lib.qh_discard_pending(dev.h))
        dev.reset_stats()
    except Exception:  # pylint: disable=broad-except
        dev.close()
        return
    dev._parked = True            # (DeviceState.__del__ leaves parked states alone)
    _pool.setdefault((dev.nbits, dev.bit_width), []).append(dev)
    _pool_order.append(dev)
    while len(_pool_order) > keep:
        old = _pool_order.pop(0)
        _pool[(old.nbits, old.bit_width)].remove(old)
        old._parked = False
        old.close()


def make_device_state(nbits, bit_width):
    if _device_factory is not None:
        return _device_factory(nbits, bit_width)
    parked = _pool.get((nbits, bit_width))
    while parked:
        dev = parked.pop()
        _pool_order.remove(dev)
        dev._parked = False
        if getattr(dev, 'h', None):
            return dev        # (its contents are whatever the last circuit left: the caller initialises the state)
    if _pool_order and nbits > _pool_limits()[1]:
        drop_device_pool()    # a state too large to park wants the memory for its own second buffer (relayout sweeps)
    try:
        return _default_device_factory(nbits, bit_width)
    except native.QhError as e:
        if e.code != native.QH_ERR_NOMEM or not _pool_order:
            raise
        drop_device_pool()    # the parked buffers were in the way
        return _default_device_factory(nbits, bit_width)


def acquire_device_state(nbits, bit_width):
    """A device state for a State's mirror (qcc_amd.lib.state): from the pool if one of that shape is parked."""
    return make_device_state(nbits, bit_width)


_state_mirror_forced = None


def state_mirror_setting():
    return _state_mirror_forced


def set_state_mirror(on):
    """Opt in (True) to / out (False) of device mirrors for States driven through State.apply1 / applyc directly
    (qcc_amd/lib/state.py; `with state.device_mirror():` wraps this); None = the rule of state_mirror_allowed().  Switching
    the mode off brings every live mirror home first."""
    global _state_mirror_forced
    _state_mirror_forced = on
    if not state_mirror_allowed():
        from qcc_amd.lib import state
        state.sync_all_mirrors()


def state_mirror_allowed():
    """State.apply1 / applyc keep the literal per-call path (the reference's contract: the host array holds the result when
    the call returns) unless the caller opted in: set_state_mirror(True) / state.device_mirror(), or QCC_STATE_MIRROR=1 in
    the environment -- the latter only while the default host executor is in place (a test that installed its own executor
    wants to see every call) and the process is not one rank of a sharded job."""
    if _state_mirror_forced is not None:
        return bool(_state_mirror_forced)
    return (os.environ.get('QCC_STATE_MIRROR', '0') == '1'
            and (_host_executor is None or type(_host_executor) is HipHostExecutor)  # pylint: disable=unidiomatic-typecheck
            and int(os.environ.get('WORLD_SIZE', '1')) == 1)


def set_device_factory(factory):
    global _device_factory
    _device_factory = factory
