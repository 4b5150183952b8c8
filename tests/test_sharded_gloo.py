"""Multi-process CPU tests (gloo, world_size 2 and 4) of the multi-GPU layer:
routing of gates over shard bits, the pairwise half-shard exchange and the
logical->physical bit map of qcc_amd.sharded.ShardedState, against the
single-process oracle on the same gate stream."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from qcc_amd import gates, workloads
from tests.oracle_lib import NO_CTL


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _stream(n, seed):
  rng = np.random.default_rng(seed)
  pool = [gates.hadamard(), gates.pauli_x(), gates.pauli_y(), gates.tgate(), gates.u1(0.37),
          gates.rz(0.9), gates.vgate(), gates.ry(0.3)]
  ops, gs = [], []
  for _ in range(90):
    t = int(rng.integers(n))
    g = pool[int(rng.integers(len(pool)))]
    if rng.random() < 0.5:
      c = int((t + 1 + rng.integers(n - 1)) % n)
      ops.append((c, t))
    else:
      ops.append((NO_CTL, t))
    gs.append(np.asarray(g, dtype=np.complex128).reshape(4))
  sb = workloads.qft_stream(range(n))          # then a QFT: diag gates on shard bits, H on top qubits
  o2, g2 = sb.arrays()
  ops = np.concatenate([np.array(ops, dtype=np.int32), o2])
  g8 = np.concatenate([np.array(gs).view(np.float64).reshape(-1, 8), g2])
  return ops, g8


def _worker(rank, world, port, n, seed, out_dir, mode='alltoall'):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank))
  import torch.distributed as dist
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from qcc_amd import sharded
  from tests import fake_device

  def factory(nloc):
    return fake_device.NumpyShardEngine(nloc)
  st = sharded.ShardedState(n, engine_factory=factory, chunk_amps=8, exchange=mode)   # tiny chunks: many rounds
  ops, g8 = _stream(n, seed)
  x = 0b1011010 & ((1 << n) - 1)
  st.init_basis(x)
  st.run_stream(ops, g8)
  full = st.gather_logical()
  n2 = st.norm2_global()
  idx, p = st.argmax_global()
  if rank == 0:
    np.savez(os.path.join(out_dir, 'res.npz'), psi=full, norm2=n2, argmax=idx, p=p, exchanges=st.exchanges,
             perm=np.array(st.perm))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize('world,n,seed,mode', [(2, 6, 0, 'alltoall'), (4, 7, 1, 'alltoall'), (4, 8, 2, 'pairwise'),
                                               (4, 9, 3, 'alltoall')])
def test_sharded_equals_single_process_oracle(oracle, tmp_path, world, n, seed, mode):
  port = _free_port()
  mp.spawn(_worker, args=(world, port, n, seed, str(tmp_path), mode), nprocs=world, join=True)
  res = np.load(tmp_path / 'res.npz')
  ops, g8 = _stream(n, seed)
  want = np.zeros(1 << n, dtype=np.complex128)
  want[0b1011010 & ((1 << n) - 1)] = 1
  oracle.run_stream(want, n, ops, g8)
  assert np.max(np.abs(res['psi'] - want)) < 1e-12
  assert abs(float(res['norm2']) - 1) < 1e-12
  assert int(res['argmax']) == int(np.argmax(np.abs(want))) or abs(float(res['p']) - np.max(np.abs(want)) ** 2) < 1e-12
  assert int(res['exchanges']) >= 1                      # dense gates did hit shard bits
  assert sorted(res['perm'].tolist()) == list(range(n))  # the bit map stays a permutation


def test_qft_needs_exactly_g_exchanges(tmp_path):
  """36q/8 GPU QFT claim: only the H gates on the g top qubits exchange."""
  from qcc_amd import sharded  # noqa: F401  (import check)
  n, world = 7, 4
  port = _free_port()
  mp.spawn(_qft_only, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
  res = np.load(tmp_path / 'qft.npz')
  assert int(res['exchanges']) == 1                      # one all-to-all instead of g=2 pairwise steps
  k = np.arange(1 << n)
  want = workloads.qft_analytic(n, 0b0110101, k)
  assert np.max(np.abs(res['psi'] - want)) < 1e-12


@pytest.mark.parametrize('reps', [4, 5])
def test_repeated_qft_one_exchange_per_step(tmp_path, reps):
  """Belady / MRU choice of the evicted bit group: steady state = ONE exchange per QFT."""
  n, world = 10, 4
  port = _free_port()
  mp.spawn(_qft_repeat, args=(world, port, n, reps, str(tmp_path)), nprocs=world, join=True)
  res = np.load(tmp_path / 'rep.npz')
  assert int(res['exchanges']) <= reps + 1, int(res['exchanges'])
  assert np.max(np.abs(res['psi'] - res['want'])) < 1e-12


def _qft_repeat(rank, world, port, n, reps, out_dir):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank))
  import torch.distributed as dist
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from qcc_amd import sharded
  from tests import fake_device, oracle_lib

  def factory(nloc):
    return fake_device.NumpyShardEngine(nloc)
  st = sharded.ShardedState(n, engine_factory=factory, chunk_amps=16)
  st.min_evict_bit = 2
  st.init_basis(0b1100101)
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  if reps % 2:                           # odd: one call, the router sees the whole stream (Belady)
    st.run_stream(np.concatenate([ops] * reps), np.concatenate([g8] * reps))
  else:                                  # even: one call per QFT as bench.py does (MRU fallback)
    for _ in range(reps):
      st.run_stream(ops, g8)
  ops = np.concatenate([ops] * reps)
  g8 = np.concatenate([g8] * reps)
  full = st.gather_logical()
  if rank == 0:
    want = np.zeros(1 << n, dtype=np.complex128)
    want[0b1100101] = 1
    oracle_lib.load().run_stream(want, n, ops, g8)
    np.savez(os.path.join(out_dir, 'rep.npz'), psi=full, want=want, exchanges=st.exchanges)
  dist.barrier()
  dist.destroy_process_group()


def _qft_only(rank, world, port, n, out_dir):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank))
  import torch.distributed as dist
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from qcc_amd import sharded
  from tests import fake_device

  def factory(nloc):
    return fake_device.NumpyShardEngine(nloc)
  st = sharded.ShardedState(n, engine_factory=factory)
  st.init_basis(0b0110101)
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  st.run_stream(ops, g8)
  full = st.gather_logical()
  if rank == 0:
    np.savez(os.path.join(out_dir, 'qft.npz'), psi=full, exchanges=st.exchanges)
  dist.barrier()
  dist.destroy_process_group()


def _broken_transport_worker(rank, world, port, out_dir):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank))
  import torch.distributed as dist
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from qcc_amd import sharded
  from tests import fake_device

  class Broken(fake_device.NumpyShardEngine):
    def comm_init_custom(self, nranks, rank_, round_fn):
      if rank_ == 1:
        raise RuntimeError('no transport on this rank')
      super().comm_init_custom(nranks, rank_, round_fn)
  try:
    sharded.ShardedState(8, engine_factory=Broken)
    outcome = 'constructed'
  except sharded.TransportError as e:
    outcome = f'TransportError: {e}'
  with open(os.path.join(out_dir, f'outcome{rank}.txt'), 'w') as f:
    f.write(outcome)
  dist.barrier()
  dist.destroy_process_group()


def test_a_transport_that_fails_on_one_rank_is_an_error_on_every_rank(tmp_path):
  """a communicator that cannot be set up must be an error everywhere -- never a silent agreement on some
  other data path.  Rank 1's engine refuses; ranks 0..3 all raise TransportError, each saying where it failed."""
  world = 4
  mp.spawn(_broken_transport_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  texts = [open(tmp_path / f'outcome{r}.txt').read() for r in range(world)]
  assert all(t.startswith('TransportError') for t in texts), texts
  assert 'no transport on this rank' in texts[1] and 'another rank failed' in texts[0]


# ---- circuit.qc() on a sharded register (north_star: "every algorithm runs unmodified") ----------
def _grover_through_qc(nb, bits):
  """grover.py:124-168 as the reference script writes it, through qcc_amd.lib.circuit.qc."""
  import math
  from qcc_amd.lib import circuit, ops
  qc = circuit.qc('Grover')
  reg = qc.reg(nb, 0)
  qc.reg(1, 1)
  aux = qc.reg(nb - 1, 0)
  idx = list(range(nb))
  qc.h(list(range(nb + 1)))
  for _ in range(int(math.pi / 4 * math.sqrt(2 ** nb))):
    for i in idx:
      if bits[i] == 0:
        qc.apply1(ops.PauliX(), i, 'x')
    qc.multi_control(reg, nb, aux, ops.PauliX(), 'Phase Inversion')
    for i in idx:
      if bits[i] == 0:
        qc.apply1(ops.PauliX(), i, 'x')
    qc.h(idx); qc.x(idx)
    qc.multi_control(reg, nb, aux, ops.PauliZ(), 'Mean Inversion')
    qc.x(idx); qc.h(idx)
  return qc


def _qc_worker(rank, world, port, golden, out_dir):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank))
  import torch.distributed as dist
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from qcc_amd import sharded
  from qcc_amd.lib import backend, circuit, tensor
  from tests import fake_device

  def factory(nloc):
    return fake_device.NumpyShardEngine(nloc)
  made = []

  def device_factory(nbits, bw):
    d = sharded.ShardedDevice(nbits, bw, engine_factory=factory, chunk_amps=64)
    made.append(d)
    return d
  backend.set_device_factory(device_factory)
  tensor.set_tensor_width(128)
  g = np.load(golden)
  bits = [int(b) for b in g['marked']]
  qc = _grover_through_qc(len(bits), bits)
  maxbits, maxprob = qc.maxprob()                     # per-shard arg-max + all-gather
  n = qc.nbits
  p_anc, _ = qc.measure_bit(len(bits), 1, collapse=False)      # ancilla qubit: local bit on every rank
  p_top, _ = qc.measure_bit(0, bits[0], collapse=False)        # qubit 0: (originally) the shard bit
  a_marked = qc.ampl(*(bits + [1] + [0] * (len(bits) - 1)))
  psi = np.asarray(qc.psi).copy()
  # collapse on a shard-resident qubit, then on a local one
  p1, _ = qc.measure_bit(0, bits[0], collapse=True)
  p2, _ = qc.measure_bit(n - 1, 0, collapse=True)
  collapsed = np.asarray(qc.psi).copy()
  # a second register built as a product state on the shards (superposed factor on the shard bit)
  qc2 = circuit.qc('prod')
  qc2.qubit(0.6, 0.8)
  qc2.reg(5, 0b10110)
  qc2.qubit(1 / np.sqrt(2), 1j / np.sqrt(2))
  qc2.cx(0, 3)
  qc2.h(6)
  psi2 = np.asarray(qc2.psi).copy()
  if rank == 0:
    np.savez(os.path.join(out_dir, 'qc.npz'), psi=psi, maxbits=np.array(maxbits), maxprob=maxprob, p_anc=p_anc, p_top=p_top,
             a_marked=a_marked, p1=p1, p2=p2, collapsed=collapsed, psi2=psi2, exchanges=made[0].st.exchanges,
             sharded=len(made))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_qc_api_runs_on_a_sharded_register(oracle, golden_dir, tmp_path, world):
  """The reference's Grover script (nbits=6, 12 qubits) through circuit.qc() with the register sharded
  over 2 / 4 ranks: final state equals the reference's recorded state (g5_grover6.npz); maxprob,
  measure_bit (probability and collapse, on shard and local qubits), ampl and device-side product
  registers agree with a single-process run."""
  golden = os.path.join(golden_dir, 'g5_grover6.npz')
  port = _free_port()
  mp.spawn(_qc_worker, args=(world, port, golden, str(tmp_path)), nprocs=world, join=True)
  res = np.load(tmp_path / 'qc.npz')
  g = np.load(golden)
  want = g['final']
  bits = [int(b) for b in g['marked']]
  nb, n = len(bits), int(g['nbits'])
  assert int(res['sharded']) == 2 and int(res['exchanges']) >= 1
  assert np.max(np.abs(res['psi'] - want)) < 1e-10
  assert res['maxbits'][:nb].tolist() == bits
  p = np.abs(want) ** 2
  idx = np.arange(1 << n)
  assert abs(float(res['maxprob']) - p.max()) < 1e-12
  assert abs(float(res['p_anc']) - p[((idx >> (n - 1 - nb)) & 1) == 1].sum()) < 1e-12
  assert abs(float(res['p_top']) - p[((idx >> (n - 1)) & 1) == bits[0]].sum()) < 1e-12
  marked_index = int(''.join(map(str, bits + [1] + [0] * (nb - 1))), 2)
  assert abs(complex(res['a_marked']) - want[marked_index]) < 1e-12
  c = want.copy()
  c[((idx >> (n - 1)) & 1) != bits[0]] = 0
  c /= np.sqrt(float(res['p1']))
  c[(idx & 1) != 0] = 0
  c /= np.sqrt(float(res['p2']))
  assert np.max(np.abs(res['collapsed'] - c)) < 1e-10
  # the product register: kron of the factors, then cx(0,3), h(6)
  v = np.kron(np.kron(np.array([0.6, 0.8]), np.eye(32)[0b10110]), np.array([1, 1j]) / np.sqrt(2)).astype(np.complex128)
  oracle.applyc(v, np.array([0, 1, 1, 0], dtype=np.complex128), 7, 0, 3)
  oracle.apply1(v, np.array([1, 1, 1, -1], dtype=np.complex128) / np.sqrt(2), 7, 6)
  assert np.max(np.abs(res['psi2'] - v)) < 1e-12
