"""TEST-ONLY stand-ins for the GPU, used by the CPU ("not gpu") host-logic tests.

OracleDevice implements the interface circuit.qc needs from
qcc_amd.device.DeviceState with NumPy + the CPU oracle; OracleHostExecutor is
the host-buffer counterpart.  Installed through qcc_amd.lib.backend's test
seams; product code never imports this module.
"""
import numpy as np

from tests import oracle_lib


class OracleDevice:
  calls = 0

  def __init__(self, nbits, bit_width):
    self.nbits, self.bit_width = nbits, bit_width
    self.dtype = np.complex128 if bit_width == 128 else np.complex64
    self.psi = np.zeros(1 << nbits, dtype=self.dtype)
    self.o = oracle_lib.load()
    self.trace = []

  def close(self):
    pass

  def host_array(self):
    """stand-in for a host-mapped state (device.DeviceState.host_array): THE buffer gates work on"""
    return self.psi

  def init_basis(self, index=0):
    self.psi[:] = 0
    self.psi[index] = 1

  def init_product(self, factors):
    v = np.ones(1, dtype=np.complex128)
    for n, x in factors:
      if isinstance(x, (int, np.integer)):
        t = np.zeros(1 << n, dtype=np.complex128)
        t[int(x)] = 1
      else:
        t = np.asarray(x, dtype=np.complex128).reshape(-1)
      v = np.kron(v, t)
    self.psi[:] = v.astype(self.dtype)

  def upload(self, host, offset=0):
    host = np.asarray(host, dtype=self.dtype)
    self.psi[offset:offset + host.size] = host

  def download(self, offset=0, count=None, out=None):
    count = self.psi.size - offset if count is None else count
    if out is not None:
      out[:count] = self.psi[offset:offset + count]
      return out
    return self.psi[offset:offset + count].copy()

  def apply1(self, gate, index):
    OracleDevice.calls += 1
    self.trace.append((None, int(index), np.asarray(gate, dtype=np.complex128).reshape(4).copy()))
    self.o.apply1(self.psi, gate, self.nbits, index)

  def applyc(self, gate, control, target):
    OracleDevice.calls += 1
    self.trace.append((int(control), int(target), np.asarray(gate, dtype=np.complex128).reshape(4).copy()))
    self.o.applyc(self.psi, gate, self.nbits, control, target)

  def sync(self):
    pass

  def flush(self):
    pass

  def norm2(self):
    return float(np.vdot(self.psi, self.psi).real)

  def argmax(self):
    i = int(np.argmax(np.abs(self.psi)))
    return i, float(np.abs(self.psi[i]) ** 2)

  def prob_bit(self, bit, value=1):
    sel = ((np.arange(self.psi.size) >> bit) & 1) == value
    return float(np.vdot(self.psi[sel], self.psi[sel]).real)

  def project_bit(self, bit, value):
    sel = ((np.arange(self.psi.size) >> bit) & 1) != value
    self.psi[sel] = 0

  def scale(self, z):
    self.psi *= z

  def amplitude(self, index):
    return self.psi[index]


class PlanDevice(OracleDevice):
  """Like OracleDevice, but the queued gates go through the REAL planner: at every flush the
  queue is planned by the engine library (dry handle, qh_plan_export) and the plan is executed
  with NumPy (tests/plan_interp.py).  States below 8 qubits (no sweeps) use the oracle."""
  planned_flushes = 0

  def __init__(self, nbits, bit_width):
    super().__init__(nbits, bit_width)
    self.queue = []

  def apply1(self, gate, index):
    self.queue.append((None, int(index), np.asarray(gate, dtype=np.complex128).reshape(4).copy()))

  def applyc(self, gate, control, target):
    self.queue.append((int(control), int(target), np.asarray(gate, dtype=np.complex128).reshape(4).copy()))

  def flush(self):
    import ctypes
    from qcc_amd import native
    from tests import plan_interp
    q, self.queue = self.queue, []
    if not q:
      return
    n = self.nbits
    in_range = all(c is None or 0 <= c < n for c, _, _ in q)
    if n < 8 or not in_range:          # no sweeps / the reference's out-of-range control quirk: oracle semantics
      for c, t, g in q:
        if c is None:
          super().apply1(g, t)
        else:
          super().applyc(g, c, t)
      return
    lib = native.load()
    h = ctypes.c_void_p()
    native.check(lib.qh_create_dry(n, 128, ctypes.byref(h)))
    native.check(lib.qh_set_fusion(h, native.QH_FUSE_SWEEP))
    dp = ctypes.POINTER(ctypes.c_double)
    for c, t, g in q:
      g8 = np.ascontiguousarray(g).view(np.float64)
      if c is None:
        native.check(lib.qh_apply1(h, t, g8.ctypes.data_as(dp)))
      else:
        native.check(lib.qh_applyc(h, c, t, g8.ctypes.data_as(dp)))
    sweeps, _ = plan_interp.export_plan(h)
    lib.qh_destroy(h)
    work = self.psi.astype(np.complex128)
    plan_interp.run_plan(work, sweeps, n)
    self.psi[:] = work.astype(self.dtype)
    PlanDevice.planned_flushes += 1

  sync = flush

  def download(self, offset=0, count=None, out=None):
    self.flush()
    return super().download(offset, count, out)

  def upload(self, host, offset=0):
    self.flush()
    super().upload(host, offset)

  def init_basis(self, index=0):
    self.queue = []
    super().init_basis(index)

  def init_product(self, factors):
    self.queue = []
    super().init_product(factors)

  def norm2(self):
    self.flush()
    return super().norm2()

  def argmax(self):
    self.flush()
    return super().argmax()

  def prob_bit(self, bit, value=1):
    self.flush()
    return super().prob_bit(bit, value)

  def project_bit(self, bit, value):
    self.flush()
    super().project_bit(bit, value)

  def scale(self, z):
    self.flush()
    super().scale(z)

  def amplitude(self, index):
    self.flush()
    return super().amplitude(index)


class OracleHostExecutor:
  def __init__(self):
    self.o = oracle_lib.load()

  def apply1(self, psi, gate, nbits, tgt, bit_width=128):
    self.o.apply1(psi, gate, nbits, tgt)

  def applyc(self, psi, gate, nbits, ctl, tgt, bit_width=128):
    self.o.applyc(psi, gate, nbits, ctl, tgt)


class NumpyShardEngine:
  """CPU stand-in for the per-rank engine of qcc_amd.sharded.ShardedState, with the interface of
  qcc_amd.device.DeviceState as that layer uses it: the engine knows its shard (set_shard) and resolves shard-bit
  controls / diagonal gates on shard bits itself, builds states in place, and runs the exchange as host-staged
  ROUNDS through the callback ShardedState hands it (comm_init_custom) -- the same protocol the HIP engine speaks
  when several ranks share one GPU.  NumPy only; TEST INFRASTRUCTURE."""

  def __init__(self, nloc, bit_width=128):
    self.nbits = nloc
    self.nbits_global = nloc
    self.shard = 0
    self.bit_width = bit_width
    self.psi = np.zeros(1 << nloc, dtype=np.complex128)
    self.n_gates = 0
    self._round_fn = None
    self._x = {'exchanges': 0, 'rounds': 0, 'bytes_sent': 0, 'slabs': 0, 'sweeps_overlapped': 0, 'span_ms': 0.0, 'rounds_packed': 0}

  # -- configuration ------------------------------------------------------------------
  def set_shard(self, nbits_global, shard_index):
    self.nbits_global, self.shard = int(nbits_global), int(shard_index)

  def set_relayout(self, on):
    return False                       # one buffer: nothing to re-lay out

  # -- state construction / IO -----------------------------------------------------------
  def init_basis(self, index=0):
    self.psi[:] = 0
    if (int(index) >> self.nbits) == self.shard:
      self.psi[int(index) & ((1 << self.nbits) - 1)] = 1

  def init_product(self, factors):
    idx = (np.uint64(self.shard) << np.uint64(self.nbits)) | np.arange(1 << self.nbits, dtype=np.uint64)
    out = np.ones(idx.shape, dtype=np.complex128)
    shift = self.nbits_global
    for n, x in factors:
      shift -= n
      v = ((idx >> np.uint64(shift)) & np.uint64((1 << n) - 1)).astype(np.int64)
      if isinstance(x, (int, np.integer)):
        out *= (v == int(x))
      else:
        out *= np.asarray(x, dtype=np.complex128).reshape(-1)[v]
    self.psi[:] = out

  def upload(self, host, offset=0):
    host = np.asarray(host, dtype=np.complex128).reshape(-1)
    self.psi[offset:offset + host.size] = host

  def download(self, offset=0, count=None, out=None):
    count = self.psi.size - offset if count is None else count
    return self.psi[offset:offset + count].copy()

  # -- gates (GLOBAL bit positions: bits >= nbits are the shard index) ----------------------
  def apply_bits(self, ctl_mask, tgt_bit, gate):
    g = np.asarray(gate, dtype=np.complex128).reshape(2, 2)
    self.n_gates += 1
    nloc = self.nbits
    hi = int(ctl_mask) >> nloc
    if (self.shard & hi) != hi:
      return                                       # a control lives in the shard index and is 0 here
    cm = int(ctl_mask) & ((1 << nloc) - 1)
    idx = np.arange(self.psi.size)
    on = (idx & cm) == cm
    if tgt_bit >= nloc:                            # diagonal gate on a shard bit: a rank-dependent factor
      assert g[0, 1] == 0 and g[1, 0] == 0, 'dense gate on a shard bit: exchange first'
      f = g[1, 1] if (self.shard >> (tgt_bit - nloc)) & 1 else g[0, 0]
      self.psi[on] *= f
      return
    sel = on & (((idx >> tgt_bit) & 1) == 0)
    lo = idx[sel]
    hi_ = lo | (1 << tgt_bit)
    a, b = self.psi[lo].copy(), self.psi[hi_].copy()
    self.psi[lo] = g[0, 0] * a + g[0, 1] * b
    self.psi[hi_] = g[1, 0] * a + g[1, 1] * b

  def apply_bits_raw(self, ctl_mask, tgt_bit, addr):
    import ctypes
    g = np.ctypeslib.as_array(ctypes.cast(addr, ctypes.POINTER(ctypes.c_double)), shape=(8,)).copy()
    self.apply_bits(ctl_mask, tgt_bit, g.view(np.complex128))

  # -- the exchange: host-staged rounds (qcc_hip.h qh_round_fn semantics) ------------------------
  def comm_init_custom(self, nranks, rank, round_fn):
    self._nranks, self._rank, self._round_fn = int(nranks), int(rank), round_fn

  def comm_destroy(self):
    self._round_fn = None

  def _swap_blocks(self, moves, base, gbits, chunk_amps):
    """moves: [(peer, block value sent, block value where the peer's data lands)] of the gbits local bits at base."""
    run, nruns = 1 << base, 1 << (self.nbits - base - gbits)
    view = self.psi.reshape(nruns, 1 << gbits, run)
    send = [np.ascontiguousarray(view[:, blk, :]).reshape(-1) for _, blk, _ in moves]
    recv = [np.empty_like(s_) for s_ in send]
    chunk = max(1, int(chunk_amps) or send[0].size)
    for off in range(0, send[0].size, chunk):
      self._round_fn([p for p, _, _ in moves], [s_[off:off + chunk].view(np.uint8) for s_ in send],
                     [r_[off:off + chunk].view(np.uint8) for r_ in recv])
      self._x['rounds'] += 1
    for (_, _, land), r_ in zip(moves, recv):
      view[:, land, :] = r_.reshape(nruns, run)
    self._x['exchanges'] += 1
    self._x['bytes_sent'] += sum(s_.nbytes for s_ in send)

  def exchange_alltoall(self, base_bit, chunk_amps=0):
    g = self._nranks.bit_length() - 1
    self._swap_blocks([(j, j, j) for j in range(self._nranks) if j != self._rank], int(base_bit), g, chunk_amps)

  def exchange_pair(self, shard_bit, local_bit, chunk_amps=0):
    mybit = (self._rank >> shard_bit) & 1
    self._swap_blocks([(self._rank ^ (1 << shard_bit), 1 - mybit, 1 - mybit)], int(local_bit), 1, chunk_amps)

  def exchange_stats(self):
    return dict(self._x)

  # -- readers ---------------------------------------------------------------------------------------
  def sync(self):
    pass

  def flush(self):
    pass

  def close(self):
    pass

  def norm2(self):
    return float(np.vdot(self.psi, self.psi).real)

  def argmax(self):
    i = int(np.argmax(np.abs(self.psi)))
    return (self.shard << self.nbits) | i, float(np.abs(self.psi[i]) ** 2)

  def amplitude(self, global_index):
    assert (int(global_index) >> self.nbits) == self.shard
    return self.psi[int(global_index) & ((1 << self.nbits) - 1)]

  def prob_bit(self, bit, value=1):
    sel = ((np.arange(self.psi.size) >> bit) & 1) == value
    return float(np.vdot(self.psi[sel], self.psi[sel]).real)

  def project_bit(self, bit, value):
    self.psi[((np.arange(self.psi.size) >> bit) & 1) != value] = 0

  def scale(self, z):
    self.psi *= complex(z)

  def stats(self):
    return {'gates_submitted': self.n_gates}

  def reset_stats(self):
    self.n_gates = 0
