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
    e = fake_device.NumpyShardEngine(nloc)
    return e, e.buf
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
    e = fake_device.NumpyShardEngine(nloc)
    return e, e.buf
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
    e = fake_device.NumpyShardEngine(nloc)
    return e, e.buf
  st = sharded.ShardedState(n, engine_factory=factory)
  st.init_basis(0b0110101)
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  st.run_stream(ops, g8)
  full = st.gather_logical()
  if rank == 0:
    np.savez(os.path.join(out_dir, 'qft.npz'), psi=full, exchanges=st.exchanges)
  dist.barrier()
  dist.destroy_process_group()
