"""State vectors and registers (API mirror of /root/reference/src/lib/state.py).

``State`` is a host-side ndarray (so ``psi[i]``, ``np.abs(psi)``, slicing keep
working exactly as in the reference).  Its two hot methods, ``apply1`` and
``applyc`` (state.py:80-125, the pure-Python specification of the native loops),
run on the GPU through the C-ABI drop-in instead of Python loops.
"""
import cmath
import math
import os
import random
import weakref

import numpy as np

from qcc_amd.lib import backend
from qcc_amd.lib import helper
from qcc_amd.lib import tensor


# ---- device mirror of a State that is driven through apply1 / applyc directly (OPT-IN) ---------------------------
# The literal contract of the reference's State.apply1 / applyc (state.py:80-125) is "the host array is updated when the
# call returns"; through the C-ABI drop-in that is H2D + kernel + D2H per gate (38 ms per gate at 26 qubits).  That literal
# path is the DEFAULT.  A caller that runs a loop of direct calls on a large State (the style of grover.py:77,
# counting.py:54, sat3.py:129, minimum_finding.py:88, state_prep.py:53) can opt in to a DEVICE MIRROR:
#
#     with state.device_mirror():            # or backend.set_state_mirror(True), or QCC_STATE_MIRROR=1 in the environment
#         for ...: psi.apply1(g, i)
#
# The first call then uploads the buffer once and every call queues its gate on the device (fused sweeps); the amplitudes
# come back -- one download -- the first time anything LOOKS at the array through Python: indexing, iteration, printing,
# every NumPy function or operator (__array_ufunc__ / __array_function__), every attribute or method other than the handful
# that do not expose data, and when the block ends.  After a look the host buffer is the truth again (the caller may have
# written to it): the next apply call uploads anew.  Only States of at least QCC_STATE_MIRROR_MIN_QUBITS qubits (default 18:
# 4 MiB, where PCIe stops mattering) are mirrored.  Other State objects over the same memory (a slice taken before the gates,
# the array a slice was taken from) are covered: mirrors are registered by the bytes they stand for (_LIVE), and the hooks
# below sit on EVERY State while the mode is on (they are installed on the class when the first mirror is made and removed
# when the mode is off and no mirror is left: a process that never opts in pays nothing).
#
# What the opt-in gives up -- a SILENT difference, not a loud one: C code that reads the memory without going through the
# State object is not intercepted.  np.asarray(psi) / np.array(psi), memoryview(psi), psi.ctypes / psi.data hand-offs made
# BEFORE the gates, the plain ndarray `a` of State(a) (a State is a view of what it was made from), an extension that kept a
# pointer: while the device is ahead they all read NaN -- the host bytes are poisoned on purpose, so that such a read gives
# an unmistakably wrong array rather than a plausible state that lacks the gates, but nothing raises.  Code with such readers
# must not opt in (or must look at the State itself first: psi[:], any NumPy function, leaving the `with` block); the literal
# drop-in (qcc_amd.dropin.libxgates) and the default path keep the reference's contract byte for byte.
_mirror_totals = {'uploads': 0, 'downloads': 0, 'h2d_bytes': 0, 'd2h_bytes': 0, 'gates': 0}


def mirror_stats(reset=False):
    """PCIe traffic of the State mirrors of this process (tests, tools)."""
    out = dict(_mirror_totals)
    if reset:
        for k in _mirror_totals:
            _mirror_totals[k] = 0
    return out


def _mirror_min_qubits():
    return int(os.environ.get('QCC_STATE_MIRROR_MIN_QUBITS', '18'))


class device_mirror:   # pylint: disable=invalid-name  (used as `with state.device_mirror():`)
    """Opt in to device mirrors of directly driven States for the duration of the block; on exit every mirror goes home."""

    def __init__(self, on=True):
        self.on, self.prev = on, None

    def __enter__(self):
        self.prev = backend.state_mirror_setting()
        backend.set_state_mirror(self.on)
        return self

    def __exit__(self, *exc):
        backend.set_state_mirror(self.prev)
        return False


class _Mirror:
    __slots__ = ('dev', 'ahead', 'lo', 'hi', 'owner')

    def __init__(self, dev, lo, hi, owner):
        self.dev = dev          # the device state (qcc_amd.device.DeviceState or a substitute with its interface)
        self.ahead = False      # the device holds gates the host buffer has not seen
        self.lo, self.hi = lo, hi       # the bytes of host memory this mirror stands for
        self.owner = owner      # weakref to the State that holds it


# Mirrors alive right now, by id(mirror).  A State is an ndarray: other State objects may be views of the same memory (a slice
# taken before the gates, the array a slice was taken from).  Whoever looks at memory a live mirror stands for brings that
# mirror home first, whichever object holds it: _sync_host checks this registry (empty almost always: one dict test per look).
_LIVE = {}

try:
    from numpy.lib.array_utils import byte_bounds as _byte_bounds      # NumPy >= 2.0
except ImportError:                                                    # NumPy 1.x
    _byte_bounds = getattr(np, 'byte_bounds', None)


def _byte_range(arr):
    a = np.ndarray.view(arr, np.ndarray)
    if _byte_bounds is not None:
        return _byte_bounds(a)
    lo = hi = a.__array_interface__['data'][0]
    for dim, stride in zip(a.shape, a.strides):
        if dim == 0:
            return lo, lo
        if stride < 0:
            lo += (dim - 1) * stride
        else:
            hi += (dim - 1) * stride
    return lo, hi + a.itemsize


def _sync_overlapping(arr):
    lo, hi = _byte_range(arr)
    for m in list(_LIVE.values()):
        if m.lo < hi and lo < m.hi:
            owner = m.owner()
            if owner is not None:
                _sync_host(owner)
            else:
                _LIVE.pop(id(m), None)


def sync_all_mirrors():
    """Every live mirror goes home (backend.set_state_mirror calls this when the mode is switched off); the hooks leave
    the class with the last one."""
    for m in list(_LIVE.values()):
        owner = m.owner()
        if owner is not None:
            _sync_host(owner)
        else:
            _LIVE.pop(id(m), None)
    _remove_hooks()


def _sync_host(self):
    """Host buffer := device state, if the device is ahead; the mirror is then dropped (the host may be written to)."""
    d = object.__getattribute__(self, '__dict__')
    m = d.get('_mirror')
    if m is None:
        if _LIVE:
            _sync_overlapping(self)
        return
    d['_mirror'] = None
    _LIVE.pop(id(m), None)
    try:
        if m.ahead:
            buf = np.ndarray.view(self, np.ndarray)
            m.dev.download(out=buf)
            _mirror_totals['downloads'] += 1
            _mirror_totals['d2h_bytes'] += buf.nbytes
    finally:
        backend.release_device_state(m.dev)


# -- the mirror's guard: anything that can see the amplitudes brings them home first.  These functions are State methods
#    only while the mirror mode is on (_install_hooks / _remove_hooks): a plain State is a plain ndarray subclass. ----------
_SAFE_ATTRS = frozenset((
    'apply1', 'applyc', 'nbits', 'shape', 'dtype', 'ndim', 'size', 'itemsize', 'nbytes', 'flags', 'strides', 'name', 'basis_index',
    '_mirror', '_exec_buffer', '_sync_host', '_mirror_apply', 'base', '__class__', '__dict__', '__array_finalize__', '__array_priority__',
    '__del__', '__init__', '__new__', '__weakref__', '__doc__', '__module__', '__slots__'))
_HOOKS = {}
_hooks_on = False


def _hook(fn):
    _HOOKS[fn.__name__] = fn
    return fn


@_hook
def __getattribute__(self, name):   # noqa: N807  pylint: disable=redefined-builtin
    if name not in _SAFE_ATTRS:
        d = object.__getattribute__(self, '__dict__')
        if d.get('_mirror') is not None or _LIVE:
            _sync_host(self)
    return object.__getattribute__(self, name)


@_hook
def __del__(self):   # noqa: N807
    d = object.__getattribute__(self, '__dict__')
    m = d.get('_mirror')
    if m is None:
        return
    if m.ahead and np.ndarray.base.__get__(self) is not None:
        # gates pending on a State that does not own its memory (a view, a State made from an array: State(a) aliases a):
        # whoever holds that memory may still look at it -- the gates go home now, always (no guessing from reference
        # counts).  A State that OWNS its buffer cannot die while a view of it lives (the view holds it as its base), so
        # only such a sole holder takes its gates to the grave without a download.
        _sync_host(self)
        return
    d['_mirror'] = None
    _LIVE.pop(id(m), None)
    try:
        backend.release_device_state(m.dev)
    except Exception:  # pylint: disable=broad-except
        pass


def _synced(name):
    def method(self, *args, **kwargs):
        _sync_host(self)
        return getattr(super(State, self), name)(*args, **kwargs)
    method.__name__ = name
    return _hook(method)


for _name in ('__getitem__', '__setitem__', '__iter__', '__repr__', '__str__', '__reduce_ex__', '__copy__', '__deepcopy__',
              '__bool__', '__complex__', '__float__', '__int__', '__contains__'):
    _synced(_name)

_QUIET = {}      # State class -> the same class with ndarray's own __array_ufunc__ ("no override" to NumPy); never handed out


def _quiet_class(cls):
    q = _QUIET.get(cls)
    if q is None:
        q = _QUIET[cls] = type('_Quiet' + cls.__name__, (cls,), {'__array_ufunc__': np.ndarray.__array_ufunc__})
    return q


@_hook
def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):   # noqa: N807
    # Bring every State operand home, then let NumPy do exactly what it does for a subclass WITHOUT this hook: the operands
    # are viewed as their class's quiet sibling (ndarray's own __array_ufunc__), so result types, priorities and out=
    # handling are NumPy's; results of a quiet class are handed back as the class they stand for.  (Any subclass of State
    # gets a sibling of its own: viewing only exact States left a user subclass to recurse into this hook for ever.)
    outs = kwargs.get('out')
    outs = (outs,) if outs is not None and not isinstance(outs, tuple) else (outs or ())
    for x in inputs + tuple(outs):
        if isinstance(x, State):
            _sync_host(x)
    quiet = lambda x: np.ndarray.view(x, _quiet_class(type(x))) if isinstance(x, State) and type(x) not in _QUIET.values() else x   # noqa: E731
    if outs:
        kwargs['out'] = tuple(quiet(x) for x in outs)
    res = getattr(ufunc, method)(*[quiet(x) for x in inputs], **kwargs)

    def loud(r):
        for cls, q in _QUIET.items():
            if type(r) is q:   # pylint: disable=unidiomatic-typecheck
                for o in outs:           # an out= operand comes back as the object the caller passed
                    if o is not None and type(o) is cls and np.shares_memory(o, r) and o.shape == r.shape:   # pylint: disable=unidiomatic-typecheck
                        return o
                return np.ndarray.view(r, cls)
        return r
    return tuple(loud(r) for r in res) if isinstance(res, tuple) else loud(res)


@_hook
def __array_function__(self, func, types, args, kwargs):   # noqa: N807
    def walk(o):
        if isinstance(o, State):
            _sync_host(o)
        elif isinstance(o, (list, tuple)):
            for y in o:
                walk(y)
        elif isinstance(o, dict):
            for y in o.values():
                walk(y)
    walk(args)
    walk(kwargs)
    return super(State, self).__array_function__(func, types, args, kwargs)


def _install_hooks():
    global _hooks_on
    if not _hooks_on:
        for name, fn in _HOOKS.items():
            setattr(State, name, fn)
        _hooks_on = True


def _remove_hooks():
    global _hooks_on
    if _hooks_on and not _LIVE and not backend.state_mirror_allowed():
        for name in _HOOKS:
            delattr(State, name)
        _hooks_on = False


class State(tensor.Tensor):
    """Amplitudes of an n-qubit pure state, qubit 0 = most significant index bit."""

    def __array_finalize__(self, obj):
        super().__array_finalize__(obj)
        self.basis_index = None  # set by the basis-state constructors below
        self._mirror = None      # (a view or a copy never shares the mirror of the array it was made from)

    def _sync_host(self):
        """Host buffer := device state if a device mirror is ahead (opt-in mode only; otherwise nothing to do)."""
        _sync_host(self)

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
        return np.ndarray.view(self, np.ndarray)

    def _mirror_apply(self, gate, control, target, n, width):
        """Queue the gate on this State's device mirror (created and filled on first use).  False: take the literal path."""
        d = object.__getattribute__(self, '__dict__')
        m = d.get('_mirror')
        if m is None:
            if not backend.state_mirror_allowed():      # the default: the literal per-call path
                return False
            lo = _mirror_min_qubits()
            if lo <= 0 or n < lo:
                return False
            buf = self._exec_buffer()
            _install_hooks()
            if _LIVE:
                _sync_overlapping(self)     # (another State over the same memory holds a mirror: one mirror per byte)
            dev = backend.acquire_device_state(n, width)
            try:
                dev.upload(buf)
            except Exception:
                backend.release_device_state(dev)
                raise
            _mirror_totals['uploads'] += 1
            _mirror_totals['h2d_bytes'] += buf.nbytes
            # While the device is ahead the host bytes are POISON (NaN), not the old amplitudes: what no Python hook can see
            # -- np.asarray(psi) / np.array(psi) / memoryview hand the memory out in C without asking the subclass, and so
            # does any extension that kept a pointer -- then reads NaN (silently: nothing raises) instead of a plausible
            # state that lacks the gates.  Part of the opt-in's contract (see the head of this file).
            buf.view(np.float64 if buf.dtype == np.complex128 else np.float32).fill(np.nan)
            lo_b, hi_b = _byte_range(self)
            m = d['_mirror'] = _Mirror(dev, lo_b, hi_b, weakref.ref(self))
            _LIVE[id(m)] = m
        g = np.asarray(gate).reshape(4)
        if control is None:
            m.dev.apply1(g, target)
        else:
            m.dev.applyc(g, control, target)
        m.ahead = True
        _mirror_totals['gates'] += 1
        return True

    def apply1(self, gate, index):
        n = self.nbits
        if not 0 <= index < n:
            raise ValueError(f'apply1: qubit {index} out of range for {n} qubits')
        width = 128 if self.dtype == np.complex128 else 64
        if self._mirror_apply(gate, None, index, n, width):
            return
        backend.host_executor().apply1(self._exec_buffer(), np.asarray(gate).reshape(4), n, index, width)

    def applyc(self, gate, control, target):
        n = self.nbits
        if not 0 <= target < n:
            raise ValueError(f'applyc: qubit {target} out of range for {n} qubits')
        width = 128 if self.dtype == np.complex128 else 64
        # (out-of-range controls: the reference's quirk Q7 lives in the literal drop-in; the mirror takes ordinary gates)
        if 0 <= control < n and control != target and self._mirror_apply(gate, control, target, n, width):
            return
        _sync_host(self)
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
