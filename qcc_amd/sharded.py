"""ShardedState: one 2^n-amplitude state across P = 2^g GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on
ROCm).  Rank r holds the 2^(n-g) amplitudes whose g TOP physical index bits
equal r, in its own HBM, behind its own engine handle.  Per gate:

  * target local, controls local         -> local kernels, no communication;
  * control on a shard bit               -> ranks whose bit is 0 skip the gate,
                                            the others run it without that control;
  * DIAGONAL gate touching shard bits    -> a phase that depends on the rank:
                                            local diagonal gate / scalar, no
                                            communication (all 630 CU1 of a
                                            36-qubit QFT are in this class or local);
  * dense gate whose TARGET is a shard bit -> ONE exchange swaps all g shard bits with g
    consecutive local bits: rank r sends block j of its shard to rank j and receives block r
    of rank j's (grouped send/recv to all P-1 peers at once, so every xGMI link of the GPU
    carries 1/P of the shard; chunked through a double-buffered staging area, in place), the
    logical->physical bit map is updated, and the gate -- and every later gate on those
    qubits -- is local.  WHICH local bits leave is chosen like a cache victim from the known
    gate stream (Belady): a QFT repeated in a loop pays exactly one exchange per QFT.
    (exchange='pairwise' keeps the one-bit variant: half a shard to rank r ^ 2^k over one link.)

There is no collective on the data path other than that exchange; reductions (norm,
arg-max) are 8-16 byte all-reduces / all-gathers.

The exchange itself runs BEHIND THE C-ABI (qh_exchange_alltoall / qh_exchange_pair,
qcc_amd/csrc/exchange.hip.h): ncclSend/ncclRecv on the engine's own exchange stream, landing
copies on a third stream, and -- cut into slabs -- overlapped with the last sweep before and the
first sweep after it on the compute stream (HIP events, no host wait).  This module only routes
gates (ShardRouter: pure Python, no torch), keeps the logical->physical bit map and hands RCCL's
unique id around through torch.distributed (control plane).  ONE transport family: the engine's.
`self.exchange_path` says how its rounds travel:
  'rccl'        RCCL send/recv over xGMI (the nccl backend: one rank per GPU),
  'host-staged' the same rounds carried by a torch.distributed/gloo callback (several ranks
                sharing one GPU: tests; fabrics without peer access).
If the transport cannot be set up on EVERY rank, ShardedState raises on every rank (bench.py turns
that into a JSON line with "error"); there is no second data path to fall back to.

The local engine is a qcc_amd.device.DeviceState that OWNS its shard (qh_create + qh_set_shard): it
may then re-lay the shard out between its two buffers like a single-GPU handle does (relayout sweeps,
planner.h), and the exchange follows the index bits wherever they are (packed rounds).  Every rank
submits EVERY gate to its engine -- also the ones whose shard-bit control is 0 on this rank: the
planner keeps those as ghosts, so that all ranks plan the same sweeps, tiles, slabs and layouts (an
exchange needs them to agree; engine.hip verify_geometry checks it before data moves).  Tests
substitute an engine of the same interface (tests/fake_device.py NumpyShardEngine, host-staged rounds
over gloo) through `engine_factory` to exercise exactly this routing / bit-map code with world sizes
2 and 4; DryShard runs one rank's routing and PLANNING without any device or process group.
"""
import math
import os

import numpy as np

NO_CTL = -(2 ** 31)


# ---- what a rank needs, and what a step should cost (the first real multi-GPU line is judged against these) ----
XGMI_LINK_GBPS = 153.0          # per link and direction, peak (AMD's public figure; 7 links per GPU, one per peer in an 8-GPU node)
SWEEP_MS_AT_2P33 = 46.7         # one relayout sweep over a 2^33-amplitude complex128 shard, measured (profiles/r05/qft33_kernel_stats.csv)
PACK_GBPS = 5000.0              # gather / scatter kernels of packed rounds: bytes moved per second (k_xpack / k_xunpack stream like a sweep)


def memory_plan(nbits, world, bit_width=128, chunk_amps=1 << 22):
  """Device memory ONE rank asks for, in bytes: the shard, the second buffer relayout sweeps need, the staging halves of
  the exchange (two send + two receive halves of (P-1) chunks: exchange.hip.h).  `need_in_place` is the floor (a rank that
  cannot get the second buffer makes every rank fall back to in-place sweeps), `need_relayout` the default."""
  g = int(math.log2(world))
  nloc = int(nbits) - g
  amp = 16 if int(bit_width) == 128 else 8
  shard = amp << nloc
  chunk = min(int(chunk_amps), 1 << max(0, nloc - 1))
  staging = 4 * (world - 1) * chunk * amp if world > 1 else 0
  return {'qubits': int(nbits), 'ranks': int(world), 'local_qubits': nloc, 'shard_bytes': shard, 'second_buffer_bytes': shard,
          'staging_bytes': staging, 'need_in_place_bytes': shard + staging, 'need_relayout_bytes': 2 * shard + staging}


def predict_step_ms(nbits, world, *, sweeps=None, exchanges=None, bit_width=128, link_efficiency=0.7, overlap_efficiency=0.8):
  """Predicted time of ONE n-qubit QFT on `world` GPUs, so that a measured line can be called good or bad:
    sweeps x t_sweep            t_sweep = 46.7 ms x 2^(local qubits - 33) (measured at 2^33; complex64: half), 3 sweeps on one GPU,
                                4 with an exchange (tools/plan_sharded.py: what the planner does for every rung of the ladder)
  + per exchange  max(t_link, t_pack + hidden) - hidden
                                t_link  = shard / P bytes per link and direction (all-to-all: every peer's link carries 1/P of the
                                          shard) / (153 GB/s x link_efficiency);
                                t_pack  = the gather + scatter kernels of packed rounds: 4 x (P-1)/P x shard bytes at 5 TB/s,
                                          on the same HBM the overlapped sweeps use;
                                hidden  = overlap_efficiency x 2 x 7/8 x t_sweep: the slabs of the sweep before and of the sweep
                                          after the exchange that run while the links are busy (8 slabs).
  `expected` uses link_efficiency 0.7 / overlap 0.8, `best_case` 1.0 / 1.0.  Returns ms and the terms."""
  g = int(math.log2(world))
  nloc = int(nbits) - g
  amp = 16 if int(bit_width) == 128 else 8
  shard = amp << nloc
  if sweeps is None:
    sweeps = 3 if world == 1 else 4
  if exchanges is None:
    exchanges = 0 if world == 1 else 1
  t_sweep = SWEEP_MS_AT_2P33 * 2.0 ** (nloc - 33) * (amp / 16.0)

  def total(link_eff, ov):
    if not exchanges:
      return sweeps * t_sweep, 0.0, 0.0, 0.0
    t_link = shard / world / (XGMI_LINK_GBPS * link_eff * 1e9) * 1e3
    t_pack = 4.0 * (world - 1) / world * shard / (PACK_GBPS * 1e9) * 1e3
    hidden = ov * 2 * 7 / 8 * t_sweep
    window = max(t_link, t_pack + hidden)
    return sweeps * t_sweep + exchanges * (window - hidden), t_link, t_pack, hidden
  exp, t_link, t_pack, hidden = total(link_efficiency, overlap_efficiency)
  best = total(1.0, 1.0)[0]
  return {'expected_ms': exp, 'best_case_ms': best, 'sweeps': sweeps, 'exchanges': exchanges, 'sweep_ms': t_sweep,
          'link_ms': t_link, 'pack_ms': t_pack, 'hidden_ms': hidden,
          'assumptions': (f'{XGMI_LINK_GBPS:g} GB/s per xGMI link and direction x {link_efficiency:g}; sweep {SWEEP_MS_AT_2P33:g} ms per 2^33 '
                          f'amplitudes (measured); packed rounds at {PACK_GBPS / 1e3:g} TB/s; {overlap_efficiency:g} of 2 x 7/8 sweeps hidden')}


class MemoryPlanError(RuntimeError):
  """A rank's shard + staging does not fit its free device memory (raised on EVERY rank, before anything is allocated)."""


class ShardRouter:
  """The routing of a gate stream over the shard bits, as ONE rank does it: logical -> physical bit map, which
  gates force an exchange, which local bits give way (Belady), the bookkeeping of the swaps.  Pure Python over an
  engine object (qcc_amd.device.DeviceState or anything with its interface); no torch, no process group."""

  def __init__(self, nbits, world, rank, eng, *, exchange='alltoall', chunk_amps=1 << 22):
    self.rank, self.world = int(rank), int(world)
    self.g = int(math.log2(self.world))
    assert 1 << self.g == self.world, 'number of ranks must be a power of two'
    self.nbits = int(nbits)
    self.nloc = self.nbits - self.g
    assert self.nloc >= 2, 'shard too small'
    self.eng = eng
    # logical bit b (0 = least significant; qubit q is bit nbits-1-q) -> physical bit
    self.perm = list(range(self.nbits))
    self.chunk = min(int(chunk_amps), 1 << (self.nloc - 1))
    assert exchange in ('alltoall', 'pairwise')
    self.exchange_mode = exchange if self.nloc >= 2 * self.g else 'pairwise'
    self.min_evict_bit = max(2, min(20, self.nloc - 3 * self.g))
    self._last_use = {}      # physical bit -> sequence number of its last use as a dense target
    self._seq = 0
    self.exchanges = 0
    self.exchanged_bytes = 0
    self.gates = 0
    self._native_chunk = int(os.environ.get('QCC_EXCHANGE_CHUNK_AMPS', '0')) or self.chunk

  @property
  def amp_bytes(self):
    return 16 if getattr(self, 'bit_width', 128) == 128 else 8

  # ------------------------------------------------------------------ helpers
  def _phys_mask(self, logical_mask):
    m, b = 0, 0
    while logical_mask:
      if logical_mask & 1:
        m |= 1 << self.perm[b]
      logical_mask >>= 1
      b += 1
    return m

  def logical_to_phys(self, idx):
    return self._phys_mask(idx)

  def phys_to_logical(self, idx):
    out = 0
    for b in range(self.nbits):
      if (idx >> self.perm[b]) & 1:
        out |= 1 << b
    return out

  # ------------------------------------------------------------------ gates
  def apply_bits(self, ctl_mask, tgt_bit, gate):
    """Gate on LOGICAL bit tgt_bit under logical control mask (all ranks call this).  Shard-bit controls and
    diagonal gates on shard bits are resolved by the engine (it knows its shard: qh_set_shard), which must see
    every gate on every rank (ghosts, planner.h)."""
    g4 = np.asarray(gate, dtype=np.complex128).reshape(4)
    self.gates += 1
    pt = self.perm[tgt_bit]
    diag = g4[1] == 0 and g4[2] == 0
    if pt >= self.nloc and not diag:
      self._exchange(pt)                      # collective: before anything rank-dependent
      pt = self.perm[tgt_bit]
    if not diag:
      self._seq += 1
      self._last_use[pt] = self._seq
    self.eng.apply_bits(self._phys_mask(ctl_mask), pt, g4)

  def apply1(self, gate, index):
    self.apply_bits(0, self.nbits - 1 - int(index), gate)

  def applyc(self, gate, control, target):
    c = self.nbits - 1 - int(control)
    if not 0 <= c < self.nbits:
      raise ValueError(f'control qubit {control} out of range')
    self.apply_bits(1 << c, self.nbits - 1 - int(target), gate)

  def run_stream(self, ops, gates8):
    """Replay (ops int32[G,2], gates float64[G,8]) in reference qubit numbers.

    Same routing as apply_bits, with the per-gate Python work reduced to a few
    integer operations (every gate goes straight to qh_apply_bits with a pointer into `gates8`)."""
    ops = np.ascontiguousarray(ops, dtype=np.int32)
    g8 = np.ascontiguousarray(gates8, dtype=np.float64)
    n, nloc = self.nbits, self.nloc
    diag = ((g8[:, 2:6] == 0).all(axis=1)).tolist()
    tbits = (n - 1 - ops[:, 1]).tolist()
    cq = ops[:, 0].tolist()
    base = g8.ctypes.data
    raw = self.eng.apply_bits_raw
    perm = self.perm
    for k in range(len(cq)):
      tb = tbits[k]
      pt = perm[tb]
      if pt >= nloc and not diag[k]:
        evict = self._evict_group(tbits, diag, k) if self.exchange_mode == 'alltoall' else None
        self._exchange(pt, evict)
        perm = self.perm
        pt = perm[tb]
      if not diag[k]:
        self._seq += 1
        self._last_use[pt] = self._seq
      if cq[k] == NO_CTL:
        cm = 0
      else:
        c = n - 1 - cq[k]
        if not 0 <= c < n:
          raise ValueError(f'control qubit {cq[k]} out of range')
        if c == tb:
          raise ValueError(f'control == target (qubit {cq[k]})')
        cm = 1 << perm[c]
      raw(cm, pt, base + 64 * k)
    self.gates += len(cq)

  # ------------------------------------------------------------------ the exchange step
  def _exchange(self, shard_phys_bit, base=None):
    # asynchronous: queued sweeps, rounds and the following sweeps are ordered by HIP events
    # inside the engine; its exchange timer is a HIP-event span (stats())
    if self.exchange_mode == 'alltoall':
      base = self.nloc - self.g if base is None else int(base)
      self.eng.exchange_alltoall(base, self._native_chunk)
      self._record_all(base)
    else:
      self.eng.exchange_pair(shard_phys_bit - self.nloc, self.nloc - 1, self._native_chunk)
      self._record_pair(shard_phys_bit)

  def _evict_group(self, tbits, diag, k):
    """Which g consecutive local bits to hand to the shard index when gate k of a
    known stream forces an exchange: the aligned group (not below bit
    `min_evict_bit`, so runs stay >= 16 MiB at benchmark sizes) whose qubits are
    needed as a dense TARGET farthest in the future (Belady).  A QFT repeated in a
    loop then pays ONE exchange per QFT instead of two."""
    g, nloc = self.g, self.nloc
    lo = min(self.min_evict_bit, nloc - g)
    groups = [b for b in range(nloc - g, lo - 1, -g)]
    if len(groups) <= 1:
      return nloc - g
    where = {}                                       # physical bit -> group base
    for b in groups:
      for j in range(g):
        where[b + j] = b
    nxt = {b: None for b in groups}
    pending = len(groups)
    horizon = min(len(tbits), k + 1 + 4096)
    for m in range(k + 1, horizon):
      if diag[m]:
        continue
      gb = where.get(self.perm[tbits[m]])
      if gb is not None and nxt[gb] is None:
        nxt[gb] = m
        pending -= 1
        if pending == 0:
          break
    never = [b for b in groups if nxt[b] is None]
    if never:
      # no known future use: assume the access pattern repeats (loops over the same
      # circuit) and give away the group used MOST recently -- it is needed last
      return max(never, key=lambda b: max(self._last_use.get(b + j, -1) for j in range(g)))
    return max(groups, key=lambda b: nxt[b])

  def _record_all(self, base):
    g = self.g
    for k in range(g):                               # shard bit k <-> local bit base+k
      a_phys, b_phys = self.nloc + k, base + k
      la, lb = self.perm.index(a_phys), self.perm.index(b_phys)
      self.perm[la], self.perm[lb] = b_phys, a_phys
    self.exchanges += 1
    self.exchanged_bytes += (self.world - 1) * (1 << (self.nloc - g)) * self.amp_bytes

  def _record_pair(self, shard_phys_bit):
    top = self.nloc - 1
    la = self.perm.index(shard_phys_bit)
    lb = self.perm.index(top)
    self.perm[la], self.perm[lb] = top, shard_phys_bit
    self.exchanges += 1
    self.exchanged_bytes += (1 << top) * self.amp_bytes


class DryShard(ShardRouter):
  """ONE rank of a sharded run, planned and never executed: the routing above over a planner-only engine handle
  (qh_create_dry + qh_set_shard + qh_comm_init_dry).  `geometries` collects how each exchange would be cut
  (qh_xgeom: signature, slabs, rounds, chunk size, packed / direct, staging bytes, sweeps planned before it).  No
  device, no torch, no process group: the pre-flight check of a multi-GPU configuration --
  tests/test_exchange_geometry_cpu.py plans BASELINE config 5 (36 qubits on 8 GPUs) for all eight ranks this way
  and compares what they would do."""

  def __init__(self, nbits, world, rank, *, bit_width=128, relayout=True, **kw):
    from qcc_amd import device, native
    self.bit_width = int(bit_width)
    nloc = int(nbits) - int(math.log2(world))
    eng = device.DeviceState(nloc, bit_width, fusion=native.QH_FUSE_SWEEP, dry=True)
    eng.set_shard(nbits, rank)
    eng.comm_init_dry(world, rank)
    super().__init__(nbits, world, rank, eng, **kw)
    self.geometries = []

  def _exchange(self, shard_phys_bit, base=None):
    super()._exchange(shard_phys_bit, base)
    geo = self.eng.exchange_geometry()
    geo['bitmap_after'] = list(self.perm)
    self.geometries.append(geo)

  def flush(self):
    self.eng.flush()

  def stats(self):
    s = self.eng.stats()
    s.update(exchanges=self.exchanges, exchanged_bytes=self.exchanged_bytes)
    return s

  def close(self):
    self.eng.close()


def _hip_engine_factory(nloc, local_rank, fusion, bit_width=128):
  """The engine of one rank on cuda:local_rank; it owns its shard."""
  import torch
  from qcc_amd import device
  if not torch.cuda.is_available():
    raise RuntimeError('torch sees no GPU.  If the engine library was loaded before torch was imported, two HIP '
                       'runtimes are mapped (see qcc_amd.native._preload_torch_runtime): import torch first or '
                       'launch through torchrun / set QCC_PRELOAD_TORCH=1')
  torch.cuda.set_device(local_rank)
  return device.DeviceState(nloc, bit_width, device=local_rank, fusion=fusion)


class TransportError(RuntimeError):
  """The engine-native exchange transport could not be set up on every rank (raised on EVERY rank)."""


class ShardedState(ShardRouter):
  """State (complex128, or complex64 with bit_width=64) sharded by its top log2(P) physical index bits."""

  def __init__(self, nbits, fusion=1, local_rank=None, *, engine_factory=None, backend=None,
               chunk_amps=1 << 22, exchange='alltoall', bit_width=128):
    import torch
    import torch.distributed as dist
    self.torch, self.dist = torch, dist
    if not dist.is_initialized():
      # QCC_DIST_BACKEND=gloo: several ranks on ONE GPU (tests of this layer); RCCL refuses that
      backend = backend or os.environ.get('QCC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
      os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
      if 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:   # not under torchrun: a world of one
        os.environ.update(RANK='0', WORLD_SIZE='1')
        os.environ.setdefault('MASTER_PORT', '29533')
      kw = {}
      if backend == 'nccl' and local_rank is not None:
        torch.cuda.set_device(local_rank)
        try:
          kw['device_id'] = torch.device(f'cuda:{local_rank}')
        except Exception:  # pylint: disable=broad-except
          kw = {}
      import datetime
      # (a rank that never shows up must end the job, not hang it: bench.py turns the timeout into an error line)
      kw['timeout'] = datetime.timedelta(seconds=int(os.environ.get('QCC_DIST_TIMEOUT_S', '900')))
      dist.init_process_group(backend=backend, **kw)
    rank, world = dist.get_rank(), dist.get_world_size()
    g = int(math.log2(world))
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) if local_rank is None else local_rank
    self.bit_width = int(bit_width)
    self.cdtype = np.complex128 if self.bit_width == 128 else np.complex64
    self._local_rank = local_rank
    nloc = int(nbits) - g
    self.memory_plan = self._check_memory_plan(int(nbits), world, rank, local_rank, chunk_amps, engine_factory is None)
    factory = engine_factory or (lambda nl: _hip_engine_factory(nl, local_rank, fusion, self.bit_width))
    eng = factory(nloc)
    eng.set_shard(int(nbits), rank)     # the engine resolves shard-bit controls itself and sees every gate on every rank
    super().__init__(nbits, world, rank, eng, exchange=exchange, chunk_amps=chunk_amps)
    self.exchange_seconds = 0.0
    self.exchange_path = 'none (one rank: nothing to exchange)'
    self._x0 = {}
    if self.world > 1 or os.environ.get('QCC_EXCHANGE') == 'native':
      self._init_transport()
    self.relayout = self._agree_on_relayout()

  def _check_memory_plan(self, nbits, world, rank, local_rank, chunk_amps, real_device):
    """Before anything is allocated: what this rank will ask for (shard, second buffer, staging) against what the device
    has free (hipMemGetInfo), printed per rank on stderr; if the floor -- shard + staging, in-place sweeps -- does not fit
    on ANY rank, every rank raises MemoryPlanError (bench.py: one JSON line with "error") instead of an OOM mid-run."""
    import sys
    plan = memory_plan(nbits, world, self.bit_width, chunk_amps)
    free = total = None
    if real_device and self.torch.cuda.is_available():
      try:
        free, total = (int(v) for v in self.torch.cuda.mem_get_info(local_rank))
      except Exception:  # pylint: disable=broad-except
        free = total = None
    margin = 512 << 20                       # op buffers, tables, the runtime's own
    plan.update(free_bytes=free, total_bytes=total,
                fits_in_place=None if free is None else bool(plan['need_in_place_bytes'] + margin <= free),
                fits_relayout=None if free is None else bool(plan['need_relayout_bytes'] + margin <= free))
    gib = lambda v: 'n/a' if v is None else f'{v / 2**30:.2f} GiB'   # noqa: E731
    print(f'[qcc_amd.sharded rank {rank}/{world}] memory plan: shard {gib(plan["shard_bytes"])} + second buffer '
          f'{gib(plan["second_buffer_bytes"])} + staging {gib(plan["staging_bytes"])} = {gib(plan["need_relayout_bytes"])} '
          f'(in place: {gib(plan["need_in_place_bytes"])}); free {gib(free)} of {gib(total)}'
          + ('' if plan['fits_relayout'] in (None, True) else
             ('; no room for the second buffer: in-place sweeps on every rank' if plan['fits_in_place'] else '; DOES NOT FIT')),
          file=sys.stderr, flush=True)
    ok = 0 if plan['fits_in_place'] is False else 1
    if world > 1:
      t = self.torch.tensor([ok], dtype=self.torch.int32, device='cpu' if self.dist.get_backend() == 'gloo' else f'cuda:{local_rank}')
      self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
      everyone = int(t.item())
    else:
      everyone = ok
    if not everyone:
      raise MemoryPlanError(
          f'rank {rank}: ' + (f'shard + staging = {gib(plan["need_in_place_bytes"])} exceed the {gib(free)} free on device {local_rank}'
                              if not ok else 'fits here, another rank does not have the room')
          + f' ({nbits} qubits on {world} ranks = 2^{plan["local_qubits"]} amplitudes per rank)')
    return plan

  def _init_transport(self):
    """The engine's exchange transport: RCCL under the nccl backend, host-staged rounds carried by gloo otherwise.
    Every rank must end up with the same transport: the ranks agree on success, and a failure anywhere raises
    TransportError EVERYWHERE -- a mis-set-up communicator is an error message, never a different data path."""
    dist, torch = self.dist, self.torch
    err = None
    try:
      if dist.get_backend() == 'gloo':
        def round_fn(peers, send, recv):
          ops = []
          for p, s_, r_ in zip(peers, send, recv):
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(s_), p))
            ops.append(dist.P2POp(dist.irecv, torch.from_numpy(r_), p))
          for req in dist.batch_isend_irecv(ops):
            req.wait()
        self.eng.comm_init_custom(self.world, self.rank, round_fn)
        self.exchange_path = 'host-staged'
      else:
        # every rank runs the SAME collective sequence whatever happens on rank 0: it always broadcasts -- the id, or None
        # and why it has none (librccl missing is the likeliest setup failure) -- and everybody meets in the MIN below
        box = [None, None]
        if self.rank == 0:
          try:
            box[0] = self.eng.comm_unique_id()
          except Exception as e0:  # pylint: disable=broad-except
            box[1] = f'{type(e0).__name__}: {e0}'
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
          raise TransportError(f'rank 0 could not create the RCCL unique id: {box[1]}')
        self.eng.comm_init(self.world, self.rank, box[0])
        self.exchange_path = 'rccl'
    except Exception as e:  # pylint: disable=broad-except
      err = f'{type(e).__name__}: {e}'
      self.exchange_path = f'failed on rank {self.rank} ({err})'
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=self._red_device())
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
      if err is None:
        try:
          self.eng.comm_destroy()
        except Exception:  # pylint: disable=broad-except
          pass
        self.exchange_path = 'failed on another rank'
      raise TransportError(f'the engine-native exchange transport could not be set up on every rank: rank {self.rank}: '
                           f'{err or "ok here, another rank failed"}')

  @property
  def _native(self):
    return self.exchange_path in ('rccl', 'host-staged')

  def _agree_on_relayout(self):
    """Relayout sweeps need a second buffer of the shard's size on EVERY rank (the ranks must hold the same
    layout when they exchange): each rank tries, and one that cannot makes all of them give it back."""
    if not hasattr(self.eng, 'set_relayout'):
      return False
    if self.world == 1:
      return None                            # the engine decides at its first flush, like any single-GPU handle
    # (the memory plan first: a second buffer that fits only by taking the room of the exchange's staging halves would turn
    #  into an allocation failure at the first exchange -- such a rank votes for in-place sweeps without trying)
    fits = getattr(self, 'memory_plan', {}).get('fits_relayout')
    mine = 1 if fits is not False and self.eng.set_relayout(True) else 0
    t = self.torch.tensor([mine], dtype=self.torch.int32, device=self._red_device())
    self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
    if int(t.item()) == 0:
      self.eng.set_relayout(False)
      return False
    return True

  def init_basis(self, index):
    """|index> (logical); only the owning rank gets the 1."""
    # (this layer's physical bits are the engine's logical ones: it keeps its own map of where its relayout sweeps
    #  have moved the local bits)
    self.eng.init_basis(self.logical_to_phys(int(index)))

  # ------------------------------------------------------------------ readers
  def flush(self):
    self.eng.flush()

  def sync(self):
    self.eng.sync()

  def _red_device(self):
    """Device for the tiny reduction tensors: the shard's device under RCCL, host under gloo."""
    if self.dist.get_backend() == 'gloo':
      return 'cpu'
    return f'cuda:{self._local_rank}'

  def norm2_global(self):
    if self.exchange_path == 'rccl':          # 8 bytes over the engine's own communicator
      return float(self.eng.allreduce_sum([self.eng.norm2()])[0])
    t = self.torch.tensor([self.eng.norm2()], dtype=self.torch.float64, device=self._red_device())
    self.dist.all_reduce(t)
    return float(t.item())

  def argmax_global(self):
    """(logical index, probability) of the likeliest basis state."""
    li, p = self.eng.argmax()
    phys = (self.rank << self.nloc) | (li & ((1 << self.nloc) - 1))   # (an engine that knows its shard reports global bits)
    if self.exchange_path == 'rccl':
      # one communicator for everything the data path does: the engine's (ONE all-reduce of 2 x world doubles:
      # every rank fills its own slots; indices < 2^53 are exact in a double)
      v = np.zeros(2 * self.world)
      v[2 * self.rank], v[2 * self.rank + 1] = p, float(phys)
      v = self.eng.allreduce_sum(v)
      best = max(range(self.world), key=lambda r: (float(v[2 * r]), -r))
      return self.phys_to_logical(int(v[2 * best + 1])), float(v[2 * best])
    t = self.torch.tensor([p, float(self.rank)], dtype=self.torch.float64, device=self._red_device())
    allp = [self.torch.zeros_like(t) for _ in range(self.world)]
    self.dist.all_gather(allp, t)
    idx = self.torch.tensor([phys], dtype=self.torch.int64, device=self._red_device())
    alli = [self.torch.zeros_like(idx) for _ in range(self.world)]
    self.dist.all_gather(alli, idx)
    best = max(range(self.world), key=lambda r: (float(allp[r][0]), -r))
    return self.phys_to_logical(int(alli[best].item())), float(allp[best][0])

  def amplitude_local(self, logical_index):
    """Amplitude if this rank owns it, else None."""
    phys = self.logical_to_phys(int(logical_index))
    if (phys >> self.nloc) != self.rank:
      return None
    return self.eng.amplitude(phys)             # (the engine knows its shard: global index)

  def gather_logical(self):
    """Whole state in LOGICAL order on every rank (tests / small n only)."""
    torch = self.torch
    self.eng.sync()
    # the engine owns the shard: download it (canonical order of its local bits)
    mine = torch.from_numpy(self.eng.download().view(np.float64 if self.bit_width == 128 else np.float32))
    if self.dist.get_backend() != 'gloo':
      mine = mine.to(self._red_device())
    parts = [torch.zeros_like(mine) for _ in range(self.world)]
    self.dist.all_gather(parts, mine)
    phys = np.concatenate([p_.cpu().numpy().view(self.cdtype) for p_ in parts])
    idx = np.arange(1 << self.nbits, dtype=np.uint64)
    pidx = np.zeros_like(idx)
    for b in range(self.nbits):
      pidx |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(self.perm[b])
    return phys[pidx]

  def timer_begin(self):
    self.eng.timer_begin()

  def timer_end(self):
    return self.eng.timer_end()

  def timer_lap(self):
    if hasattr(self.eng, 'timer_lap'):
      self.eng.timer_lap()

  def timer_laps(self):
    return self.eng.timer_laps() if hasattr(self.eng, 'timer_laps') else []

  def stats(self):
    s = self.eng.stats()
    s['exchanges'] = self.exchanges
    s['exchanged_bytes'] = self.exchanged_bytes
    s['exchange_seconds'] = self.exchange_seconds
    s['exchange_path'] = self.exchange_path
    if self._native:
      x = self.eng.exchange_stats()
      s['exchange_seconds'] = (x['span_ms'] - self._x0.get('span_ms', 0.0)) * 1e-3   # HIP events, first send .. last landing
      s['exchange_rounds'] = x['rounds'] - self._x0.get('rounds', 0)
      s['exchange_slabs'] = x['slabs'] - self._x0.get('slabs', 0)
      s['sweeps_overlapped_with_exchange'] = x['sweeps_overlapped'] - self._x0.get('sweeps_overlapped', 0)
      s['exchange_geometry_checks'] = x.get('geometry_checks', 0)      # since the communicator was made (each distinct geometry once)
      s['comm_ranks_reported'] = x.get('comm_ranks', 0)                # ncclCommCount / the host-staged transport's size
      if hasattr(self.eng, 'exchange_geometry') and self.exchanges:
        s['exchange_geometry'] = self.eng.exchange_geometry()
    return s

  def reset_stats(self):
    self.eng.reset_stats()
    self.exchanges = 0
    self.exchanged_bytes = 0
    self.exchange_seconds = 0.0
    self._x0 = self.eng.exchange_stats() if self._native else {}

  def close(self):
    self.eng.sync()
    self.eng.close()


class ShardedDevice:
  """The device-state interface `circuit.qc` drives (qcc_amd.device.DeviceState's), over a
  ShardedState: with WORLD_SIZE > 1 every rank runs the same Python program (the reference's
  algorithms are plain scripts: src/lib/circuit.py:68-101 and its callers) and each call below is
  collective.  Readers return the same value on every rank.

  Reference counterparts: state construction circuit.py:121-164, maxprob state.py:60-78,
  measure_bit circuit.py:287-300 -- here as per-shard device reductions plus an 8-byte all-reduce."""

  def __init__(self, nbits, bit_width=128, fusion=1, **kw):
    self.st = ShardedState(nbits, fusion=fusion, bit_width=bit_width, **kw)
    self.nbits, self.bit_width = int(nbits), int(bit_width)
    self.dtype = self.st.cdtype

  # -- helpers ---------------------------------------------------------------------
  def _all_sum(self, values):
    st = self.st
    if st.exchange_path == 'rccl':           # the engine's communicator (one communicator on the data path)
      return [float(v) for v in st.eng.allreduce_sum([float(v) for v in values])]
    t = st.torch.tensor([float(v) for v in values], dtype=st.torch.float64, device=st._red_device())
    st.dist.all_reduce(t)
    return [float(v) for v in t.tolist()]

  def _reset_map(self):
    self.st.eng.sync()
    self.st.perm = list(range(self.st.nbits))

  # -- initialisation / IO ---------------------------------------------------------
  def init_basis(self, index=0):
    self._reset_map()
    self.st.init_basis(index)

  def init_product(self, factors):
    """Each rank builds ITS slice of f_0 (x) f_1 (x) ... in place (qh_init_product knows the shard)."""
    self._reset_map()
    self.st.eng.init_product(factors)

  def upload(self, host, offset=0):
    assert offset == 0
    st = self.st
    self._reset_map()
    a = np.ascontiguousarray(host, dtype=self.dtype).reshape(-1)
    st.eng.upload(a[st.rank << st.nloc: (st.rank + 1) << st.nloc])

  def download(self, offset=0, count=None, out=None):
    full = self.st.gather_logical()
    count = full.size - offset if count is None else count
    return full[offset: offset + count]

  # -- gates -------------------------------------------------------------------------
  def apply1(self, gate, index):
    self.st.apply1(gate, index)

  def applyc(self, gate, control, target):
    self.st.applyc(gate, control, target)

  def run_stream(self, ops, gates8):
    self.st.run_stream(ops, gates8)

  def flush(self):
    self.st.flush()

  def sync(self):
    self.st.sync()

  # -- readers ---------------------------------------------------------------------
  def norm2(self):
    return self.st.norm2_global()

  def argmax(self):
    return self.st.argmax_global()

  def amplitude(self, logical_index):
    a = self.st.amplitude_local(logical_index)
    re, im = self._all_sum([0.0, 0.0] if a is None else [a.real, a.imag])
    return complex(re, im)

  def prob_bit(self, logical_bit, value=1):
    st = self.st
    pb = st.perm[int(logical_bit)]
    value = 1 if value else 0
    if pb >= st.nloc:
      p = st.eng.norm2() if ((st.rank >> (pb - st.nloc)) & 1) == value else 0.0
    else:
      p = st.eng.prob_bit(pb, value)
    return self._all_sum([p])[0]

  def project_bit(self, logical_bit, value):
    st = self.st
    pb = st.perm[int(logical_bit)]
    value = 1 if value else 0
    if pb >= st.nloc:
      if ((st.rank >> (pb - st.nloc)) & 1) != value:
        st.eng.scale(0.0)
    else:
      st.eng.project_bit(pb, value)

  def scale(self, z):
    self.st.eng.scale(z)

  def stats(self):
    return self.st.stats()

  def close(self):
    self.st.close()
