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
import sys
import weakref

import numpy as np

from qcc_amd.lib import backend
from qcc_amd.lib import helper
from qcc_amd.lib import tensor


# ---- device mirror of a State that is driven through apply1 / applyc directly ------------------------------------
# The literal contract of the reference's State.apply1 / applyc (state.py:80-125) is "the host array is updated when the
# call returns"; through the C-ABI drop-in that is H2D + kernel + D2H per gate (38 ms per gate at 26 qubits).  A loop of
# direct calls (grover.py:77, counting.py:54, sat3.py:129, minimum_finding.py:88, state_prep.py:53 and anything users
# write in that style) instead gets a DEVICE MIRROR: the first call uploads the buffer once and every call queues its
# gate on the device (fused sweeps); the amplitudes come back -- one download -- the first time anything LOOKS at the
# array: indexing, iteration, printing, every NumPy function or operator (__array_ufunc__ / __array_function__), every
# attribute or method other than the handful that do not expose data.  After such a look the host buffer is the truth
# again (the caller may have written to it): the next apply call uploads anew.  States below QCC_STATE_MIRROR_MIN_QUBITS
# (default 18: 4 MiB, where PCIe stops mattering) and every State while a test has installed its own host executor keep
# the literal per-call path; QCC_STATE_MIRROR_MIN_QUBITS=0 switches the mirror off.  Other State objects over the same memory
# (a slice taken before the gates, the array a slice was taken from) are covered: mirrors are registered by the bytes they
# stand for (_LIVE).  What no hook can intercept is C code reading the memory directly -- np.asarray(psi) / np.array(psi) do
# not consult an ndarray subclass, an extension may have kept a pointer --: while the device is ahead the host bytes are
# NaN, so such a read fails loudly instead of returning the state without its gates; look at the State itself (psi[:],
# any NumPy function or method) or use the literal drop-in (qcc_amd.dropin.libxgates) for such callers.
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


def _byte_range(arr):
    from numpy.lib.array_utils import byte_bounds
    return byte_bounds(np.ndarray.view(arr, np.ndarray))


def _sync_overlapping(arr):
    lo, hi = _byte_range(arr)
    for m in list(_LIVE.values()):
        if m.lo < hi and lo < m.hi:
            owner = m.owner()
            if owner is not None:
                owner._sync_host()
            else:
                _LIVE.pop(id(m), None)


_SAFE_ATTRS = frozenset((
    'apply1', 'applyc', 'nbits', 'shape', 'dtype', 'ndim', 'size', 'itemsize', 'nbytes', 'flags', 'strides', 'name', 'basis_index',
    '_mirror', '_exec_buffer', '_sync_host', '_mirror_apply', 'base', '__class__', '__dict__', '__array_finalize__', '__array_priority__',
    '__del__', '__init__', '__new__', '__weakref__', '__doc__', '__module__', '__slots__'))


class State(tensor.Tensor):
    """Amplitudes of an n-qubit pure state, qubit 0 = most significant index bit."""

    def __array_finalize__(self, obj):
        super().__array_finalize__(obj)
        self.basis_index = None  # set by the basis-state constructors below
        self._mirror = None      # (a view or a copy never shares the mirror of the array it was made from)

    # -- the mirror's guard: anything that can see the amplitudes brings them home first --------------------------
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

    def __getattribute__(self, name):
        if name not in _SAFE_ATTRS:
            d = object.__getattribute__(self, '__dict__')
            if d.get('_mirror') is not None or _LIVE:
                object.__getattribute__(self, '_sync_host')()
        return object.__getattribute__(self, name)

    def __del__(self):
        d = object.__getattribute__(self, '__dict__')
        m = d.get('_mirror')
        if m is not None and m.ahead:
            # gates pending: if anybody else can still reach the memory (the array a view was taken from, another view),
            # they go home now; the sole holder of its buffer takes them to the grave -- no download
            owner = np.ndarray.view(self, np.ndarray).base
            if owner is not None and sys.getrefcount(owner) > 3:
                object.__getattribute__(self, '_sync_host')()
                m = None
        if m is not None:          # nobody can look any more: no download
            d['_mirror'] = None
            _LIVE.pop(id(m), None)
            try:
                backend.release_device_state(m.dev)
            except Exception:  # pylint: disable=broad-except
                pass

    def __getitem__(self, key):
        self._sync_host()
        return super().__getitem__(key)

    def __setitem__(self, key, value):
        self._sync_host()
        super().__setitem__(key, value)

    def __iter__(self):
        self._sync_host()
        return super().__iter__()

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        # Bring every State operand home, then let NumPy do exactly what it does for a subclass WITHOUT this hook: the
        # operands are viewed as _QuietState (State, with ndarray's own __array_ufunc__: "no override" to NumPy), so result
        # types, priorities and out= handling are NumPy's; results of that class are handed back as State.
        outs = kwargs.get('out')
        outs = (outs,) if outs is not None and not isinstance(outs, tuple) else (outs or ())
        for x in inputs + tuple(outs):
            if isinstance(x, State):
                x._sync_host()
        quiet = lambda x: np.ndarray.view(x, _QuietState) if type(x) is State else x   # noqa: E731  pylint: disable=unidiomatic-typecheck
        if outs:
            kwargs['out'] = tuple(quiet(x) for x in outs)
        res = getattr(ufunc, method)(*[quiet(x) for x in inputs], **kwargs)

        def loud(r):
            if type(r) is _QuietState:   # pylint: disable=unidiomatic-typecheck
                for o in outs:           # an out= operand comes back as the object the caller passed
                    if o is not None and type(o) is State and np.shares_memory(o, r) and o.shape == r.shape:   # pylint: disable=unidiomatic-typecheck
                        return o
                return np.ndarray.view(r, State)
            return r
        return tuple(loud(r) for r in res) if isinstance(res, tuple) else loud(res)

    def __array_function__(self, func, types, args, kwargs):
        def walk(o):
            if isinstance(o, State):
                o._sync_host()
            elif isinstance(o, (list, tuple)):
                for y in o:
                    walk(y)
            elif isinstance(o, dict):
                for y in o.values():
                    walk(y)
        walk(args)
        walk(kwargs)
        return super().__array_function__(func, types, args, kwargs)

    def __repr__(self):
        self._sync_host()
        return super().__repr__()

    def __str__(self):
        self._sync_host()
        return super().__str__()

    def __reduce_ex__(self, protocol):
        self._sync_host()
        return super().__reduce_ex__(protocol)

    def __copy__(self):
        self._sync_host()
        return super().__copy__()

    def __deepcopy__(self, memo):
        self._sync_host()
        return super().__deepcopy__(memo)

    def __bool__(self):
        self._sync_host()
        return super().__bool__()

    def __complex__(self):
        self._sync_host()
        return super().__complex__()

    def __float__(self):
        self._sync_host()
        return super().__float__()

    def __int__(self):
        self._sync_host()
        return super().__int__()

    def __contains__(self, item):
        self._sync_host()
        return super().__contains__(item)

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
            lo = _mirror_min_qubits()
            if lo <= 0 or n < lo or not backend.state_mirror_allowed():
                return False
            buf = self._exec_buffer()
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
            # -- np.asarray(psi) / np.array(psi) hand the memory out in C without asking the subclass, and so does any
            # extension that kept a pointer -- then reads NaN, loudly, instead of a plausible state that lacks the gates.
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
        self._sync_host()
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


class _QuietState(State):
    """State as NumPy's ufunc machinery sees it inside State.__array_ufunc__: no override (never handed out)."""
    __array_ufunc__ = np.ndarray.__array_ufunc__


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
