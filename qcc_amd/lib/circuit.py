"""class qc -- the circuit front-end, with the state resident in MI355X HBM.

API mirror of /root/reference/src/lib/circuit.py (class qc :68-534): same method
names, argument order and meaning, so the reference's algorithms run unchanged.
The difference is below ``apply1``/``applyc``: instead of calling the
``libxgates`` CPU extension on a NumPy buffer per gate (circuit.py:196,213) the
gates are queued on a device state (qcc_amd.device.DeviceState -> C-ABI ->
fused HIP sweeps) and the host only sees amplitudes when it asks for them.

Host-visible state contract:
  * ``qc.psi`` returns a READ-ONLY host snapshot (a ``state.State``); it is
    downloaded lazily, only when the device state changed since the last read.
    A snapshot taken earlier is never mutated by later gates.
  * ``qc.psi = some_state`` replaces the state (uploaded before the next gate).
  * readers that do not need all amplitudes (maxprob, prob/ampl of one basis
    state, measure_bit, norm) have device-side equivalents on ``qc`` that move
    a few bytes instead of 2^n amplitudes: ``qc.maxprob()``, ``qc.ampl()``,
    ``qc.prob()``, ``qc.measure_bit()``.
  * registers created with ``reg``/``zeros``/``ones``/``bitstring`` on a circuit
    that is still a basis state never materialise 2^n amplitudes on the host
    (the reference does: circuit.py:121-129 -> np.kron).
  * ``qc(..., alias_psi=True)`` (or QCC_ALIAS_PSI=1), registers up to 26 qubits: the reference's contract
    LITERALLY -- the state lives in pinned host memory the GPU works on in place (qh_create_host_mapped),
    ``qc.psi`` is a writable State over that very memory, every gate is complete when its call returns,
    and ``p = qc.psi; qc.h(0)`` changes ``p`` (src/lib/xgates.cc:37-38).  Every gate crosses PCIe and is
    waited for: for code that depends on the aliasing, not for speed.
"""
import math
import weakref

import numpy as np

from qcc_amd import gates as _gates
from qcc_amd.lib import backend
from qcc_amd.lib import helper
from qcc_amd.lib import ir
from qcc_amd.lib import ops
from qcc_amd.lib import state
from qcc_amd.lib import tensor

try:
    from absl import flags as _flags
    for _name in ('libq', 'qasm', 'cirq', 'text', 'latex'):
        try:
            _flags.DEFINE_string(_name, '', f'Generate {_name} output file, or empty')
        except Exception:  # pylint: disable=broad-except
            pass
except Exception:  # pylint: disable=broad-except
    _flags = None

_SNAPSHOT_LIMIT_BITS = 31  # above this a full host snapshot is refused (>= 32 GiB)
_MEASURE_SNAPSHOT_BITS = 20  # measure_bit returns the real State up to here (16 MiB per call), a lazy handle above: loops of
                             # repeated measurements at 22-26 qubits would otherwise download up to 1 GiB per iteration
_QUEUE_FLUSH_GATES = 4096    # eager gates queued on the host side are handed to the engine in batches of this size, so the GPU starts
                             # working while a long circuit is still being written down
_ALIAS_LIMIT_BITS = 26     # alias_psi: registers up to this size live in host-mapped memory
_NO_CTL = -(2 ** 31)       # "no control" in a gate stream (include/qcc_hip.h QH_NO_CTL)

_U1_CACHE = {}


def _u1_operator(value):
    """ops.U1(value), built once per angle and width (a QFT asks for the same n - 1 angles n / 2 times each: two Operator
    constructions per gate were 40 % of what `qc.qft` costs in Python).  Read-only: the queue copies what it keeps."""
    key = (float(value), tensor.tensor_width())
    op = _U1_CACHE.get(key)
    if op is None:
        if len(_U1_CACHE) > 4096:
            _U1_CACHE.clear()
        op = ops.U1(value)
        op.flags.writeable = False
        _U1_CACHE[key] = op
    return op



def _dump_flags_set():
    if _flags is None:
        return False
    try:
        f = _flags.FLAGS
        return bool(f.libq + f.qasm + f.cirq + f.text + f.latex)
    except Exception:  # pylint: disable=broad-except
        return False


def _sqrt2x2(u):
    """Principal square root of a 2x2 matrix, closed form (the reference calls
    scipy.linalg.sqrtm per Toffoli, circuit.py:238; same matrix to ~1e-16)."""
    u = np.asarray(u, dtype=np.complex128)
    s = np.sqrt(u[0, 0] * u[1, 1] - u[0, 1] * u[1, 0])
    t = np.sqrt(u[0, 0] + u[1, 1] + 2 * s)
    if abs(t) < 1e-12:
        s = -s
        t = np.sqrt(u[0, 0] + u[1, 1] + 2 * s)
    return (u + s * np.eye(2)) / t


class _LazyPsi(np.lib.mixins.NDArrayOperatorsMixin):
    """What measure_bit() hands back as the state of a LARGE register (circuit.py:287-297 returns
    (prob, psi)): most callers only want the probability, so nothing is copied from the device until the
    caller actually looks.  The first look takes a snapshot (a real, read-only State) and every later
    look sees that same snapshot -- the value does not drift when more gates follow.  It is a snapshot of
    the state AT THE TIME OF THE FIRST LOOK; registers of <= 26 qubits get the real State at measurement
    time instead (see measure_bit).  Arithmetic, comparisons and ufuncs work as on the State."""

    def __init__(self, owner):
        self._owner = owner
        self._snap = None

    def _get(self):
        if self._snap is None:
            self._snap = self._owner.psi
            self._owner = None
        return self._snap

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self._get())
        return a.astype(dtype) if dtype is not None else a

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(x._get() if isinstance(x, _LazyPsi) else x for x in inputs)
        if 'out' in kwargs:
            kwargs['out'] = tuple(x._get() if isinstance(x, _LazyPsi) else x for x in kwargs['out'])
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __matmul__(self, other):
        return self._get() @ (other._get() if isinstance(other, _LazyPsi) else other)

    def __rmatmul__(self, other):
        return other @ self._get()

    def __getattr__(self, name):
        return getattr(self._get(), name)

    def __getitem__(self, key):
        return self._get()[key]

    def __len__(self):
        return len(self._get())

    def __iter__(self):
        return iter(self._get())

    def __repr__(self):
        return repr(self._get())


def _alive(ref):
    circ = ref()
    if circ is None:      # (a gate method kept beyond its circuit: `f = qc.h; del qc; f(0)`)
        raise ReferenceError('this gate method belongs to a circuit that has been released')
    return circ


class qc:
    """State + gate application + (optional) IR recording."""

    def __init__(self, name=None, eager=True, alias_psi=None):
        self.name = name
        import os as _os
        self._alias = bool(_os.environ.get('QCC_ALIAS_PSI') == '1' if alias_psi is None else alias_psi)
        self.ir = ir.Ir()
        self.eager = eager
        self.build_ir = not eager
        self.global_reg = 0
        self.sub_circuits = 0
        # state bookkeeping: exactly one of (_factors, _host, _dev) is authoritative
        self._nbits = 0
        # while no gate has run the state is a tensor product of the pieces handed to
        # reg()/qubit()/bitstring()/state(): [(nqubits, basis index | amplitudes)], most
        # significant first.  It is built ON THE DEVICE (qh_init_product) when the first
        # gate arrives -- the reference np.kron's on the host at every call
        # (circuit.py:121-123).
        self._factors = []
        self._product_flag = True  # (_is_product) True until something non-trivial happens
        self._host = None        # State snapshot (valid iff _host_ok)
        self._host_ok = False
        self._dev = None         # device state (valid iff _dev_ok)
        self._dev_ok = False
        # eager gates wait HERE, as (control | NO_CTL, target) + matrix, until something reads the state: they then
        # reach the engine through ONE native call (qh_apply_stream) instead of one FFI round trip per gate
        # (ctypes + NumPy conversions cost ~25 us per gate, the reference's own Python overhead is 6.5 us: SURVEY 8a A5)
        self._q_ops = []
        self._q_gates = []
        if _dump_flags_set():
            self.eager = False

        for gname, gate in (('h', ops.Hadamard()), ('s', ops.Sgate()), ('t', ops.Tgate()),
                            ('v', ops.Vgate()), ('x', ops.PauliX()), ('y', ops.PauliY()),
                            ('z', ops.PauliZ()), ('yroot', ops.Yroot())):
            self.add_single(gname, gate)
            self.add_single(gname + 'dag', gate.adjoint())
            self.add_ctl('c' + gname, gate)
            self.add_ctl('c' + gname + 'dag', gate.adjoint())

    # ------------------------------------------------------------------ state plumbing
    @property
    def _is_product(self):
        """Still the tensor product of the pieces handed in?  (Asking submits the gates queued on the host side
        first: whoever asks is about to read or rebuild the state.)"""
        if self._q_ops:
            self._drain()
        return self._product_flag

    @_is_product.setter
    def _is_product(self, value):
        self._product_flag = bool(value)

    @property
    def nbits(self):
        return self._nbits

    def _width(self):
        return tensor.tensor_width()

    def _ensure_device(self):
        """Make the device copy authoritative-capable and current (queued gates submitted)."""
        dev = self._device_ready()
        if self._q_ops:
            self._drain()
        return dev

    def _drain(self):
        """Hand the gates queued on the host side to the engine, in order."""
        if not self._q_ops:
            return
        dev = self._device_ready()        # (may raise -- no memory, width change above the snapshot limit: the queue is kept)
        ops_, gs = self._q_ops, self._q_gates
        self._q_ops, self._q_gates = [], []
        # from here on the engine owns the gates: whatever happens, neither the product description nor a host snapshot
        # describes the state any more
        self._product_flag = False
        self._host_ok = False
        if hasattr(dev, 'run_stream'):
            dev.run_stream(np.array(ops_, dtype=np.int32).reshape(-1, 2),
                           np.array(gs, dtype=np.complex128).reshape(-1, 4).view(np.float64).reshape(-1, 8))
        else:
            for (c, t), g in zip(ops_, gs):
                if c == _NO_CTL:
                    dev.apply1(g, t)
                else:
                    dev.applyc(g, c, t)
        self._gate_done()

    def flush(self):
        """Submit everything queued so far (host side and engine side) without waiting for it."""
        if self._nbits:
            dev = self._ensure_device()
            if hasattr(dev, 'flush'):
                dev.flush()

    def _device_ready(self):
        if self._nbits == 0:
            raise ValueError('circuit has no qubits yet')
        if self._dev is not None and (self._dev.nbits != self._nbits or self._dev.bit_width != self._width()):
            if (self._dev_ok and not self._host_ok and not self._product_flag and self._dev.nbits == self._nbits):
                # the tensor width changed under a live state (the reference silently stops applying gates then,
                # SURVEY appendix B Q1): carry the amplitudes over through the host
                if self._nbits > _SNAPSHOT_LIMIT_BITS:
                    raise ValueError(f'tensor width changed to {self._width()} while a {self._nbits}-qubit state is '
                                     f'resident at width {self._dev.bit_width}')
                self._host, self._host_ok = state.State(self._dev.download()), True
            if not self._alias:
                backend.release_device_state(self._dev)     # (alias mode: State views handed out keep the mapped memory alive)
            self._dev, self._dev_ok = None, False
        if self._dev is None:
            if self._alias and self._nbits <= _ALIAS_LIMIT_BITS:
                self._dev = backend.make_host_mapped_state(self._nbits, self._width())
            else:
                self._dev = backend.make_device_state(self._nbits, self._width())
        if not self._dev_ok:
            if self._product_flag:          # (the raw flag: queued gates run right after, on this very state)
                self._dev.init_product(self._factors)
            else:
                assert self._host_ok, 'no valid copy of the state'
                self._dev.upload(np.asarray(tensor.host_current(self._host)))
            self._dev_ok = True
            if self._aliased():
                self._host, self._host_ok = None, False   # from now on THE state is the mapped buffer
        return self._dev

    def _aliased(self):
        return self._alias and self._nbits and self._nbits <= _ALIAS_LIMIT_BITS

    @property
    def psi(self):
        if self._nbits == 0:
            return state.State(1.0)
        if self._q_ops:
            self._drain()
        if self._aliased():
            dev = self._ensure_device()
            dev.sync()
            if self._host is None or not self._host_ok:
                view = state.State(dev.host_array())      # zero copy: the mapped memory itself
                view._keepalive = dev                     # pylint: disable=protected-access
                self._host, self._host_ok = view, True
            return self._host
        if not self._host_ok:
            if self._nbits > _SNAPSHOT_LIMIT_BITS:
                raise MemoryError(f'qc.psi would copy 2^{self._nbits} amplitudes to the host; use qc.maxprob(), '
                                  'qc.ampl(), qc.prob(), qc.measure_bit() (device-side readers) instead')
            if self._dev_ok:
                host = state.State(self._dev.download())
            else:  # still a product state that never reached the device
                vec = np.ones(1, dtype=tensor.tensor_type())
                for n, x in self._factors:
                    if isinstance(x, (int, np.integer)):
                        t = np.zeros(1 << n, dtype=tensor.tensor_type())
                        t[int(x)] = 1
                    else:
                        t = np.asarray(x, dtype=tensor.tensor_type())
                    vec = np.kron(vec, t)
                host = state.State(vec)
            host.flags.writeable = False
            self._host, self._host_ok = host, True
        return self._host

    @psi.setter
    def psi(self, value):
        if self._q_ops:
            self._drain()
        host = tensor.host_current(value) if isinstance(value, state.State) else state.State(value)
        if host.dtype != tensor.tensor_type():
            host = state.State(host)
        if (self._aliased() and self._dev is not None and self._dev_ok and host.ndim and host.nbits == self._nbits
                and self._dev.bit_width == self._width()):
            self._dev.sync()
            np.copyto(self._dev.host_array(), np.asarray(host))     # same register: the mapped buffer stays THE state
            self._is_product = False
            return
        self._nbits = host.nbits if host.ndim else 0
        self._host, self._host_ok = host, True
        self._dev_ok = False
        self._is_product = False
        self._factors = []

    def _tprod(self, new_state, nqubits):
        """psi <- psi (x) new_state (circuit.py:121-123)."""
        if self._is_product:
            idx = getattr(new_state, 'basis_index', None)
            self._append_factor(nqubits, idx if idx is not None else np.array(tensor.host_current(new_state), dtype=np.complex128))
            return
        cur = self.psi if self._nbits else state.State(1.0)
        self.psi = cur * new_state
        self.global_reg += nqubits

    def _append_factor(self, nqubits, what):
        self._factors.append((int(nqubits), what))
        self._nbits += nqubits
        self.global_reg += nqubits
        self._host_ok = self._dev_ok = False

    def _tprod_basis(self, nqubits, index):
        self._append_factor(nqubits, int(index))

    class scope:
        """Context manager grouping gates into a named IR section."""

        def __init__(self, ir_param, desc):
            self.ir, self.desc = ir_param, desc

        def __enter__(self):
            self.ir.section(self.desc)

        def __exit__(self, t, value, traceback):
            self.ir.end_section()

    # ------------------------------------------------------------------ state builders
    def reg(self, size, it=0, *, name=None):
        ret = state.Reg(size, it, self.global_reg)
        if self._is_product:
            self._tprod_basis(size, helper.bits2val(ret.val))
        else:
            self._tprod(ret.psi(), size)
        self.ir.reg(size, name, ret)
        return ret

    def qubit(self, alpha=None, beta=None):
        self._tprod(state.qubit(alpha, beta), 1)

    def zeros(self, n):
        self._tprod(state.zeros(n), n) if not self._is_product else self._tprod_basis(n, 0)

    def ones(self, n):
        self._tprod(state.ones(n), n) if not self._is_product else self._tprod_basis(n, 2 ** n - 1)

    def bitstring(self, *bits):
        if self._is_product:
            arr = np.asarray(bits)
            assert len(arr) and ((arr == 1) | (arr == 0)).all(), 'Bits must be 0 or 1'
            self._tprod_basis(len(bits), helper.bits2val(bits))
        else:
            self._tprod(state.bitstring(*bits), len(bits))

    def rand_bits(self, n):
        self._tprod(state.rand_bits(n), n)

    def arange(self, n):
        self.psi = state.State([float(i) for i in range(2 ** n)])
        self.global_reg += n

    def random(self, n=1):
        from scipy.stats import unitary_group  # pylint: disable=import-outside-toplevel
        self.psi = ops.Operator(unitary_group.rvs(1 << n))(state.zeros(n))

    def state(self, t):
        psi = state.State(t)
        ret = state.Reg(t.nbits, 0, self.global_reg)
        self._tprod(psi, psi.nbits)
        self.ir.reg(t.nbits, 'state', ret)
        return ret

    @staticmethod
    def _ctl_by_0(ctl):
        """[q] means "controlled by |0>" (circuit.py:166-169)."""
        if isinstance(ctl, (int, np.integer)):
            return int(ctl), False
        return ctl[0], True

    # ------------------------------------------------------------------ gates
    # The generated gate methods are instance attributes (as in the reference, circuit.py:97-101).  They reach the circuit
    # through a WEAK reference: a closure over `self` stored on `self` is a cycle, and a circuit in a cycle keeps its
    # device state -- two 16-GiB buffers at 30 qubits -- until the cycle collector happens to run; `del qc` (or the name
    # going out of scope) must give the state back to the pool at once (backend.release_device_state).
    def add_single(self, name, gate):
        ref = weakref.ref(self)
        setattr(self, name, lambda idx, cond=True: _alive(ref).apply1(gate, idx, name) if cond else None)

    def add_ctl(self, name, gate):
        ref = weakref.ref(self)
        setattr(self, name, lambda idx0, idx1, cond=True: _alive(ref).applyc(gate, idx0, idx1, name) if cond else None)

    def apply1(self, gate, idx_set, name=None, *, val=None):
        """Apply a single-qubit gate to one index or to each index of a list/Reg."""
        if isinstance(idx_set, (int, np.integer)):
            indices = [int(idx_set)]
        elif isinstance(idx_set, (state.Reg, list)):
            indices = list(idx_set)
        else:
            indices = []  # the reference silently ignores other types (quirk Q10)
        for idx in indices:
            if self.build_ir:
                self.ir.single(name, idx, gate, val)
            if self.eager:
                assert idx < self._nbits, 'Invalid qubit index'
                if self._aliased():
                    self._ensure_device().apply1(np.asarray(gate).reshape(4), idx)
                else:
                    if idx < 0:
                        raise ValueError(f'apply1: qubit {idx} out of range for {self._nbits} qubits')
                    self._q_ops.append((_NO_CTL, int(idx)))
                    self._q_gates.append(np.array(gate, dtype=np.complex128).reshape(4))
                    if len(self._q_ops) >= _QUEUE_FLUSH_GATES:
                        self._drain()
                    continue
                self._gate_done()

    def applyc(self, gate, ctl, idx, name=None, *, val=None):
        """Apply `gate` on `idx` controlled by `ctl` ([ctl] = controlled by |0>)."""
        if isinstance(idx, state.Reg):
            assert len(idx) == 1, 'Controlled n-qbit register not supported'
            idx = idx[0]
        ctl_qubit, by_0 = self._ctl_by_0(ctl)
        self.x(ctl_qubit, by_0)
        if self.build_ir:
            self.ir.controlled(name, ctl_qubit, idx, gate, val)
        if self.eager:
            assert idx < self._nbits, 'Invalid qubit index'
            if self._aliased():
                self._ensure_device().applyc(np.asarray(gate).reshape(4), ctl_qubit, idx)
            else:
                if idx < 0 or ctl_qubit >= self._nbits:
                    raise ValueError(f'applyc: qubits ({ctl_qubit}, {idx}) out of range for {self._nbits} qubits')
                if ctl_qubit == idx:
                    raise ValueError(f'applyc: control == target (qubit {idx})')
                self._q_ops.append((int(ctl_qubit), int(idx)))
                self._q_gates.append(np.array(gate, dtype=np.complex128).reshape(4))
                if len(self._q_ops) >= _QUEUE_FLUSH_GATES:
                    self._drain()
            if self._aliased():
                self._gate_done()
        self.x(ctl_qubit, by_0)

    def _gate_done(self):
        self._is_product = False
        if self._aliased():
            self._dev.sync()          # the reference's calls are synchronous: holders of qc.psi see the gate now
        else:
            self._host_ok = False

    def cx0(self, idx0, idx1):
        xgate = ops.PauliX()
        self.apply1(xgate, idx0, 'x')
        self.applyc(ops.PauliX(), idx0, idx1, 'cx')
        self.apply1(xgate, idx0, 'x')

    def cu(self, idx0, idx1, op, desc=None):
        assert op.shape[0] == 2, 'cu only supports 2x2 operators'
        self.applyc(op, idx0, idx1, desc)

    def ccu(self, idx0, idx1, idx2, op, desc=''):
        """Doubly-controlled U by the Sleator-Weinfurter construction (circuit.py:227-246)."""
        i0, c0_by_0 = self._ctl_by_0(idx0)
        i1, c1_by_0 = self._ctl_by_0(idx1)
        with self.scope(self.ir, f'CC{op.name}\\{desc}({idx0},{idx1},{idx2})'):
            self.x(i0, c0_by_0)
            self.x(i1, c1_by_0)
            v = ops.Operator(_sqrt2x2(op))
            self.cu(i0, idx2, v, (op.name or '') + '^{1/2}')
            self.cx(i0, i1)
            self.cu(i1, idx2, v.adjoint(), (op.name or '') + '^t')
            self.cx(i0, i1)
            self.cu(i1, idx2, v, (op.name or '') + '^{1/2}')
            self.x(i1, c1_by_0)
            self.x(i0, c0_by_0)

    def ccx(self, idx0, idx1, idx2):
        self.ccu(idx0, idx1, idx2, ops.PauliX(), 'ccx')

    def toffoli(self, idx0, idx1, idx2):
        self.ccu(idx0, idx1, idx2, ops.PauliX(), 'ccx')

    def u1(self, idx, val):
        self.apply1(ops.U1(val), idx, 'u1', val=val)

    def cu1(self, idx0, idx1, value):
        self.applyc(_u1_operator(value), idx0, idx1, 'cu1', val=value)

    def ccu1(self, idx0, idx1, tgt, value):
        self.ccu(idx0, idx1, tgt, ops.U1(value))

    def rx(self, idx, theta):
        self.apply1(ops.RotationX(theta), idx, 'rx', val=theta)

    def ry(self, idx, theta):
        self.apply1(ops.RotationY(theta), idx, 'ry', val=theta)

    def rz(self, idx, theta):
        self.apply1(ops.RotationZ(theta), idx, 'rz', val=theta)

    def crx(self, ctl, idx, theta):
        self.applyc(ops.RotationX(theta), ctl, idx, 'crx', val=theta)

    def cry(self, ctl, idx, theta):
        self.applyc(ops.RotationY(theta), ctl, idx, 'cry', val=theta)

    def crz(self, ctl, idx, theta):
        self.applyc(ops.RotationZ(theta), ctl, idx, 'crz', val=theta)

    def unitary(self, op, idx):
        """Arbitrary multi-qubit unitary via the full matrix (host, small n only)."""
        self.psi = ops.Operator(op)(self.psi, idx)

    # ------------------------------------------------------------------ readers / measurement
    def maxprob(self):
        """(bits, probability) of the likeliest basis state, reduced on the device."""
        if self._q_ops:
            self._drain()
        if not self._dev_ok:
            return self.psi.maxprob()
        idx, p = self._dev.argmax()
        return helper.val2bits(idx, self._nbits), p

    def ampl(self, *bits):
        if self._q_ops:
            self._drain()
        if not self._dev_ok:
            return self.psi.ampl(*bits)
        return self._dev.amplitude(helper.bits2val(bits))

    def prob(self, *bits):
        a = self.ampl(*bits)
        return np.real(np.conj(a) * a)

    def norm2(self):
        return self._ensure_device().norm2()

    def measure_bit(self, idx, tostate=0, collapse=True):
        """P(qubit idx == tostate); with collapse, project and renormalise in place.

        Device-side reduction + projection kernels (the reference builds a 4^n
        density matrix: ops.py:426-460)."""
        dev = self._ensure_device()
        bit = self._nbits - 1 - idx
        prob = dev.prob_bit(bit, 1 if tostate else 0)
        if collapse:
            assert prob > 1e-20, 'Measurement collapses to 0.0.'
            dev.project_bit(bit, 1 if tostate else 0)
            dev.scale(1.0 / math.sqrt(prob))
            self._gate_done()
        # circuit.py:291-297 returns the State itself: small registers (and aliased ones, whose psi is the
        # device's own memory) get exactly that, isinstance(psi, State) included; larger ones a lazy handle
        if self._nbits <= _MEASURE_SNAPSHOT_BITS or self._aliased():
            return prob, self.psi
        return prob, _LazyPsi(self)

    def pauli_expectation(self, idx):
        p0, _ = self.measure_bit(idx, 0, False)
        return p0 - (1 - p0)

    # ------------------------------------------------------------------ composite gates
    def swap(self, idx0, idx1):
        # pylint: disable=arguments-out-of-order
        with self.scope(self.ir, f'swap({idx0}, {idx1})'):
            self.cx(idx1, idx0)
            self.cx(idx0, idx1)
            self.cx(idx1, idx0)

    def cswap(self, ctl, idx0, idx1):
        with self.scope(self.ir, f'cswap({ctl}, {idx0}, {idx1})'):
            self.cx(idx1, idx0)
            self.ccx(ctl, idx0, idx1)
            self.cx(idx1, idx0)

    def qft(self, reg, with_swaps=False):
        """Quantum Fourier transform over `reg` (gate order of circuit.py:320-328)."""
        for i in reversed(range(len(reg))):
            self.h(reg[i])
            for j in reversed(range(i)):
                self.cu1(reg[i], reg[j], np.pi / 2 ** (i - j))
        if with_swaps:
            self.flip(reg)

    def inverse_qft(self, reg, with_swaps=False):
        if with_swaps:
            self.flip(reg)
        last = len(reg) - 1
        for pos, r in enumerate(reg):
            self.h(r)
            if pos != last:
                for y in range(pos, -1, -1):
                    self.cu1(reg[pos + 1], reg[y], -np.pi / 2 ** (pos + 1 - y))

    def multi_control(self, ctl, idx1, aux, gate, desc=''):
        """Gate on idx1 controlled by all of ctl, using len(ctl)-1 ancillae (circuit.py:341-392)."""
        if aux:
            assert len(aux) >= len(ctl) - 1, 'Incorrect number of ancilla qubits.'
        with self.scope(self.ir, f'multi-{gate.name}({ctl}, {idx1}) # {desc})'):
            if not ctl:
                self.apply1(gate, idx1, desc)
                return
            ctl = list(ctl)
            if len(ctl) == 1:
                self.applyc(gate, ctl[0], idx1, desc)
                return
            if len(ctl) == 2:
                self.ccu(ctl[0], ctl[1], idx1, gate, desc)
                return
            # AND the controls into the ancilla ladder, fire, then uncompute
            self.ccx(ctl[0], ctl[1], aux[0])
            top = 0
            for c in ctl[2:]:
                self.ccx(c, aux[top], aux[top + 1])
                top += 1
            self.applyc(gate, aux[top], idx1, desc)
            for c in reversed(ctl[2:]):
                top -= 1
                self.ccx(c, aux[top], aux[top + 1])
            self.ccx(ctl[0], ctl[1], aux[0])

    def flip(self, reg):
        for i in range(len(reg) // 2):
            self.swap(reg[i], reg[len(reg) - 1 - i])

    # ------------------------------------------------------------------ circuits of circuits
    def qc(self, qc_parm, offset=0):
        """Replay another circuit's IR on this circuit, shifted by `offset`."""
        for node in qc_parm.ir.gates:
            if node.is_single():
                self.apply1(node.gate, node.idx0 + offset, node.name, val=node.val)
            if node.is_ctl():
                self.applyc(node.gate, node.ctl + offset, node.idx1 + offset, node.name, val=node.val)

    def run(self):
        """Execute the recorded IR now (without recording it again)."""
        saved = self.build_ir, self.eager
        self.build_ir, self.eager = False, True
        self.qc(self)
        self.build_ir, self.eager = saved

    def inverse(self):
        """A new non-eager circuit with the adjoint gates in reverse order."""
        newqc = qc(self.name, eager=False)
        for node in reversed(self.ir.gates):
            val = -node.val if node.val else None
            if node.is_single():
                newqc.apply1(node.gate.adjoint(), node.idx0, node.name + '*', val=val)
            if node.is_ctl():
                newqc.applyc(node.gate.adjoint(), node.ctl, node.idx1, node.name + '*', val=val)
        return newqc

    def control_by(self, ctl):
        """Make every recorded gate additionally controlled by qubit `ctl`."""
        assert not self.eager, 'control_by() used in non-eager circuit.'
        res = ir.Ir()
        for node in self.ir.gates:
            if node.is_single():
                node.to_ctl(ctl)
                res.add_node(node)
            elif node.is_ctl():
                sub = qc('multi', eager=False)
                sub.multi_control([ctl, node.ctl], node.idx1, None, node.gate, node.desc)
                for inner in sub.ir.gates:
                    res.add_node(inner)
        self.ir = res

    def sub(self, name=''):
        made = qc(f'inner_{self.sub_circuits}{name}', eager=False)
        self.sub_circuits += 1
        return made

    # ------------------------------------------------------------------ debug / output
    def stats(self):
        return f'Circuit Statistics\n  Qubits: {self.nbits}\n  Gates : {self.ir.ngates}\n'

    def dump_to_file(self):
        """Text emitters (qasm/libq/cirq/latex) are outside the accelerated path."""
        if _dump_flags_set():
            raise NotImplementedError('circuit text dumpers are not part of qcc_amd (hot path only)')

    def dump(self, *, desc=None, draw=False, pstate=True):
        if desc:
            print(desc)
        if self.name:
            print(f'Circuit: {self.name}, Gates: {len(self.ir.gates)}, QBits: {self.nbits}')
        print(self.ir, end='')
        if pstate:
            self.psi.dump('Current state')

    def sync(self):
        """Wait for all queued device work (for timing)."""
        if self._q_ops:
            self._drain()
        if self._dev is not None:
            self._dev.sync()

    def close(self):
        """Give the device state back (to the per-process pool of qcc_amd.lib.backend: the next circuit of the same
        shape reuses its buffers instead of allocating 2 x 16 GiB again).  Also runs when the circuit is collected."""
        self._q_ops, self._q_gates = [], []
        if self._dev is not None:
            if not self._alias:
                backend.release_device_state(self._dev)
            self._dev, self._dev_ok = None, False

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass
