"""DeviceState: a 2^n amplitude vector resident in MI355X HBM behind the C-ABI.

Host-side mirror of the native half of the reference boundary: the methods
apply1/applyc take the same arguments, in the same order and meaning, as
State.apply1 / State.applyc (src/lib/state.py:80,102) and the libxgates entry
points (src/lib/xgates.cc:89-145): reference (big-endian) qubit numbers and a
2x2 (or flattened) complex gate.
"""
import ctypes

import numpy as np

from qcc_amd import gates as _gates
from qcc_amd import native

_dp = ctypes.POINTER(ctypes.c_double)


def _g8(gate):
  g = _gates.as8(gate)
  return g, g.ctypes.data_as(_dp)


def merge_factors(factors, table_bits=12, max_factors=32):
  """Adjacent factors are merged on the host while that is cheap (basis with basis;
  tables up to 2^table_bits amplitudes), so that the device sees few factors."""
  out = []
  for n, x in factors:
    n = int(n)
    if n == 0:
      continue
    is_int = isinstance(x, (int, np.integer))
    if out:
      pn, px = out[-1]
      p_int = isinstance(px, (int, np.integer))
      if is_int and p_int and pn + n <= 63:
        out[-1] = (pn + n, (int(px) << n) | int(x))
        continue
      if not (is_int and p_int) and pn + n <= table_bits:
        def table(m, v):
          if isinstance(v, (int, np.integer)):
            t = np.zeros(1 << m, dtype=np.complex128)
            t[int(v)] = 1
            return t
          return np.asarray(v, dtype=np.complex128).reshape(-1)
        out[-1] = (pn + n, np.kron(table(pn, px), table(n, x)))
        continue
    out.append((n, int(x) if is_int else np.asarray(x, dtype=np.complex128).reshape(-1)))
  while len(out) > max_factors:      # pathological: many large tables; merge the smallest neighbours
    sizes = [out[i][0] + out[i + 1][0] for i in range(len(out) - 1)]
    i = int(np.argmin(sizes))
    (n0, x0), (n1, x1) = out[i], out[i + 1]
    def tab(m, v):
      if isinstance(v, (int, np.integer)):
        t = np.zeros(1 << m, dtype=np.complex128)
        t[int(v)] = 1
        return t
      return v
    out[i:i + 2] = [(n0 + n1, np.kron(tab(n0, x0), tab(n1, x1)))]
  return out


class DeviceState:
  """Owns a qh_handle.  complex128 (bit_width=128) or complex64 (64)."""

  def __init__(self, nbits, bit_width=128, device=0, fusion=native.QH_FUSE_OFF, *,
               device_ptr=None, stream=None, host_mapped=False, dry=False):
    self.lib = native.load()
    self.nbits = int(nbits)
    self.bit_width = int(bit_width)
    self.dtype = np.complex128 if bit_width == 128 else np.complex64
    h = ctypes.c_void_p()
    if dry:        # planner-only handle (qh_create_dry): gates queue, plans and exchange geometry can be inspected, no device
      native.check(self.lib.qh_create_dry(self.nbits, self.bit_width, ctypes.byref(h)))
    elif host_mapped:
      native.check(self.lib.qh_create_host_mapped(self.nbits, self.bit_width, device, ctypes.byref(h)))
    elif device_ptr is None:
      rc = self.lib.qh_create(self.nbits, self.bit_width, device, ctypes.byref(h))
      if rc == native.QH_ERR_NOMEM:
        # device states parked by finished circuits (qcc_amd.lib.backend's pool) may be in the way: free them, try again
        from qcc_amd.lib import backend  # pylint: disable=import-outside-toplevel
        backend.drop_device_pool()
        rc = self.lib.qh_create(self.nbits, self.bit_width, device, ctypes.byref(h))
      native.check(rc)
    else:
      native.check(self.lib.qh_attach(self.nbits, self.bit_width, device,
                                      ctypes.c_void_p(device_ptr),
                                      ctypes.c_void_p(stream or 0), ctypes.byref(h)))
    self.h = h
    self.nbits_global = self.nbits
    if fusion != native.QH_FUSE_OFF:
      self.set_fusion(fusion)

  # -- lifetime ---------------------------------------------------------------
  def close(self):
    if getattr(self, 'h', None):
      self.lib.qh_destroy(self.h)
      self.h = None

  def __del__(self):
    # A state parked in qcc_amd.lib.backend's pool belongs to the pool: when a circuit and its state are collected in one
    # pass of the cycle collector, BOTH finalizers run -- the circuit's parks the state (and so resurrects it), this one must
    # then leave the handle alone.  The pool closes what it evicts or drops explicitly.
    if getattr(self, '_parked', False):
      return
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def __enter__(self):
    return self

  def __exit__(self, *a):
    self.close()

  # -- configuration ------------------------------------------------------------
  def set_fusion(self, level):
    native.check(self.lib.qh_set_fusion(self.h, int(level)))

  def set_relayout(self, on):
    """Relayout sweeps on/off (qh_set_relayout); returns the resulting mode."""
    actual = ctypes.c_int(0)
    native.check(self.lib.qh_set_relayout(self.h, 1 if on else 0, ctypes.byref(actual)))
    return bool(actual.value)

  def set_shard(self, nbits_global, shard_index):
    native.check(self.lib.qh_set_shard(self.h, int(nbits_global), int(shard_index)))
    self.nbits_global = int(nbits_global)

  def host_array(self):
    """NumPy view of a host-mapped state (qh_create_host_mapped): the very memory the GPU works on.
    Call sync() before reading."""
    p = ctypes.c_void_p()
    native.check(self.lib.qh_host_ptr(self.h, ctypes.byref(p)))
    if not p.value:
      raise ValueError('this state lives in HBM (create it with host_mapped=True)')
    real = ctypes.c_double if self.bit_width == 128 else ctypes.c_float
    flat = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(real)), shape=(2 << self.nbits,))
    return flat.view(self.dtype)

  @property
  def device_ptr(self):
    p = ctypes.c_void_p()
    native.check(self.lib.qh_device_ptr(self.h, ctypes.byref(p)))
    return p.value

  # -- initialisation / IO ------------------------------------------------------
  def init_basis(self, index=0):
    native.check(self.lib.qh_init_basis(self.h, int(index)))

  def init_product(self, factors):
    """State := f_0 (x) f_1 (x) ... built on the device.  factors: [(nqubits, x)], x an
    int (basis state |x>) or 2^nqubits amplitudes; f_0 holds the most significant qubits."""
    factors = merge_factors(factors)
    k = len(factors)
    nq = (ctypes.c_int32 * k)(*[int(f[0]) for f in factors])
    basis = (ctypes.c_uint64 * k)()
    ptrs = (ctypes.c_void_p * k)()
    keep = []
    for j, (n, x) in enumerate(factors):
      if isinstance(x, (int, np.integer)):
        basis[j] = int(x)
      else:
        a = np.ascontiguousarray(x, dtype=np.complex128).reshape(-1)
        if a.size != 1 << n:
          raise ValueError(f'factor {j}: {a.size} amplitudes for {n} qubits')
        keep.append(a)
        ptrs[j] = a.ctypes.data
    native.check(self.lib.qh_init_product(self.h, k, nq, ptrs, basis))

  def upload(self, host, offset=0):
    a = np.ascontiguousarray(host, dtype=self.dtype)
    native.check(self.lib.qh_upload(self.h, a.ctypes.data, int(offset), a.size))

  def download(self, offset=0, count=None, out=None):
    count = (1 << self.nbits) - offset if count is None else count
    if out is None:
      out = np.empty(count, dtype=self.dtype)
    assert out.dtype == self.dtype and out.flags.c_contiguous and out.size >= count
    native.check(self.lib.qh_download(self.h, out.ctypes.data, int(offset), int(count)))
    return out

  # -- the hot path ---------------------------------------------------------------
  def apply1(self, gate, index):
    _keep, p = _g8(gate)
    native.check(self.lib.qh_apply1(self.h, int(index), p))

  def applyc(self, gate, control, target):
    _keep, p = _g8(gate)
    native.check(self.lib.qh_applyc(self.h, int(control), int(target), p))

  def apply_bits(self, ctl_mask, tgt_bit, gate):
    _keep, p = _g8(gate)
    native.check(self.lib.qh_apply_bits(self.h, int(ctl_mask), int(tgt_bit), p))

  def apply_bits_raw(self, ctl_mask, tgt_bit, addr):
    """apply_bits with the gate given as the address of 8 contiguous doubles."""
    rc = self.lib.qh_apply_bits(self.h, ctl_mask, tgt_bit, ctypes.cast(addr, _dp))
    if rc:
      native.check(rc)

  def run_stream(self, ops, gates8):
    """ops int32[G,2] (ctl or NO_CTL, tgt), gates8 float64[G,8] -- reference qubit numbers."""
    ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 2)
    gates8 = np.ascontiguousarray(gates8, dtype=np.float64).reshape(-1, 8)
    assert len(ops) == len(gates8)
    # one FFI call for the stream: exactly len(ops) qh_apply1 / qh_applyc calls inside (include/qcc_hip.h)
    native.check(self.lib.qh_apply_stream(self.h, len(ops), ops.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                          gates8.ctypes.data_as(_dp)))

  def flush(self):
    native.check(self.lib.qh_flush(self.h))

  def sync(self):
    native.check(self.lib.qh_sync(self.h))

  # -- multi-GPU exchange (include/qcc_hip.h: qh_comm_* / qh_exchange_*) ------------
  @staticmethod
  def comm_unique_id():
    buf = ctypes.create_string_buffer(128)
    native.check(native.load().qh_comm_unique_id(buf))
    return buf.raw

  def comm_init(self, nranks, rank, unique_id):
    """RCCL transport: send/recv over xGMI on the engine's exchange stream."""
    assert len(unique_id) == 128
    native.check(self.lib.qh_comm_init(self.h, int(nranks), int(rank), ctypes.c_char_p(unique_id)))

  def comm_init_custom(self, nranks, rank, round_fn):
    """Host-staged transport: round_fn(peers, send_arrays, recv_arrays) moves one round (uint8 views of
    the engine's pinned buffers); used with gloo when several ranks share one GPU (tests)."""
    def _cb(_user, npeers, peers, send, recv, nbytes):
      try:
        ps = [peers[i] for i in range(npeers)]
        sv = [np.ctypeslib.as_array(ctypes.cast(send[i], ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)) for i in range(npeers)]
        rv = [np.ctypeslib.as_array(ctypes.cast(recv[i], ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)) for i in range(npeers)]
        round_fn(ps, sv, rv)
        return 0
      except Exception:  # pylint: disable=broad-except
        import traceback
        traceback.print_exc()
        return 1
    self._round_cb = native.ROUND_FN(_cb)     # keep the trampoline alive as long as the handle
    native.check(self.lib.qh_comm_init_custom(self.h, int(nranks), int(rank), self._round_cb, None))

  def comm_destroy(self):
    native.check(self.lib.qh_comm_destroy(self.h))

  def exchange_alltoall(self, base_bit, chunk_amps=0):
    native.check(self.lib.qh_exchange_alltoall(self.h, int(base_bit), int(chunk_amps)))

  def exchange_pair(self, shard_bit, local_bit, chunk_amps=0):
    native.check(self.lib.qh_exchange_pair(self.h, int(shard_bit), int(local_bit), int(chunk_amps)))

  def exchange_loopback(self, local_bit, chunk_amps=0):
    native.check(self.lib.qh_exchange_loopback(self.h, int(local_bit), int(chunk_amps)))

  def exchange_wait(self):
    native.check(self.lib.qh_exchange_wait(self.h))

  def comm_init_dry(self, nranks, rank):
    """Planner-only handles: qh_exchange_* decide slabs / rounds / path like rank `rank` of `nranks` would; nothing moves."""
    native.check(self.lib.qh_comm_init_dry(self.h, int(nranks), int(rank)))

  def exchange_geometry(self):
    """How the last exchange was cut (qh_xgeom): equal on every rank by construction, compared before data moves."""
    g = native.QhXGeom()
    native.check(self.lib.qh_exchange_geometry(self.h, ctypes.byref(g)))
    return g.as_dict()

  def exchange_stats(self):
    s = native.QhXStats()
    native.check(self.lib.qh_exchange_stats(self.h, ctypes.byref(s)))
    return s.as_dict()

  def allreduce_sum(self, values):
    a = np.ascontiguousarray(values, dtype=np.float64).copy()
    native.check(self.lib.qh_comm_allreduce_sum(self.h, a.ctypes.data_as(_dp), a.size))
    return a

  # -- readers --------------------------------------------------------------------
  def norm2(self):
    v = ctypes.c_double()
    native.check(self.lib.qh_norm2(self.h, ctypes.byref(v)))
    return v.value

  def argmax(self):
    i, p = ctypes.c_uint64(), ctypes.c_double()
    native.check(self.lib.qh_argmax(self.h, ctypes.byref(i), ctypes.byref(p)))
    return self.phys_to_logical(i.value), p.value

  def prob_bit(self, logical_bit, value=1):
    v = ctypes.c_double()
    native.check(self.lib.qh_prob_bit_value(self.h, int(logical_bit), int(value), ctypes.byref(v)))
    return v.value

  def scale(self, z):
    z = complex(z)
    native.check(self.lib.qh_scale(self.h, z.real, z.imag))

  def project_bit(self, logical_bit, value):
    native.check(self.lib.qh_project_bit(self.h, int(logical_bit), int(value)))

  def phys_to_logical(self, i):
    o = ctypes.c_uint64()
    native.check(self.lib.qh_phys_to_logical(self.h, int(i), ctypes.byref(o)))
    return o.value

  def logical_to_phys(self, i):
    o = ctypes.c_uint64()
    native.check(self.lib.qh_logical_to_phys(self.h, int(i), ctypes.byref(o)))
    return o.value

  def remap_swap(self, a, b):
    native.check(self.lib.qh_remap_swap(self.h, int(a), int(b)))

  def amplitude(self, logical_index):
    """One amplitude by LOGICAL (local) index -- 16-byte D2H."""
    out = (ctypes.c_double * 2)()
    native.check(self.lib.qh_amplitude(self.h, int(logical_index), out))
    return self.dtype(complex(out[0], out[1]))

  # -- engine measurement ---------------------------------------------------------
  def stats(self):
    s = native.QhStats()
    native.check(self.lib.qh_get_stats(self.h, ctypes.byref(s)))
    return s.as_dict()

  def reset_stats(self):
    native.check(self.lib.qh_reset_stats(self.h))

  def timer_begin(self):
    native.check(self.lib.qh_timer_begin(self.h))

  def timer_end(self):
    ms = ctypes.c_float()
    native.check(self.lib.qh_timer_end(self.h, ctypes.byref(ms)))
    return ms.value

  def timer_lap(self):
    native.check(self.lib.qh_timer_lap(self.h))

  def timer_laps(self, cap=4096):
    """Milliseconds between consecutive timer_lap() marks (waits for the last one)."""
    buf = (ctypes.c_float * cap)()
    n = ctypes.c_int(0)
    native.check(self.lib.qh_timer_laps(self.h, buf, cap, ctypes.byref(n)))
    return [float(buf[k]) for k in range(min(n.value, cap))]

  def plan_json(self):
    need = ctypes.c_uint64()
    native.check(self.lib.qh_plan_json(self.h, None, 0, ctypes.byref(need)))
    buf = ctypes.create_string_buffer(need.value)
    native.check(self.lib.qh_plan_json(self.h, buf, need.value, None))
    return buf.value.decode()


NO_CTL = -(2 ** 31)
