"""The multi-GPU layer with REAL engines: two processes share the one MI355X of
the test box (gloo between them, chunks staged through the host), each owning
half of the state in HBM through its own engine handle attached to a torch CUDA
tensor.  Exercises what the CPU gloo tests cannot: torch <-> engine memory
sharing, one HIP runtime per process, fused sweeps on shards, the exchange on
GPU buffers.  (RCCL itself needs >= 2 GPUs: run by the driver's scaling bench.)"""
import os
import socket

import numpy as np
import pytest

from qcc_amd import gates, workloads
from tests.oracle_lib import NO_CTL

pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _stream(n, seed):
  rng = np.random.default_rng(seed)
  pool = [gates.hadamard(), gates.pauli_x(), gates.tgate(), gates.u1(0.37), gates.rz(0.9), gates.vgate(),
          gates.ry(0.3), gates.yroot()]
  ops, gs = [], []
  for _ in range(60):
    t = int(rng.integers(n))
    g = pool[int(rng.integers(len(pool)))]
    if rng.random() < 0.5:
      ops.append((int((t + 1 + rng.integers(n - 1)) % n), t))
    else:
      ops.append((NO_CTL, t))
    gs.append(np.asarray(g, dtype=np.complex128).reshape(4))
  o2, g2 = workloads.qft_stream(range(n)).arrays()
  return (np.concatenate([np.array(ops, dtype=np.int32), o2]),
          np.concatenate([np.array(gs).view(np.float64).reshape(-1, 8), g2]))


def _variant_stream(n, g, seed):
  """A circuit that makes the ranks execute DIFFERENT gate lists right before an exchange:
  dense gates on high local qubits under shard-bit controls, X under a shard-bit control, diagonal gates on shard
  bits -- then dense gates on the shard qubits (exchange), more of the same, and a QFT."""
  rng = np.random.default_rng(seed)
  ops, gs = [], []
  def add(c, t, gate):
    ops.append((NO_CTL if c is None else c, t))
    gs.append(np.asarray(gate, dtype=np.complex128).reshape(4))
  for q in range(n):
    add(None, q, gates.hadamard())
  for rep_ in range(3):
    for q in range(g, n):
      r = rng.random()
      if r < 0.35:
        add(int(rng.integers(g)), q, [gates.vgate(), gates.yroot(), gates.pauli_x(), gates.ry(0.3)][int(rng.integers(4))])
      elif r < 0.5:
        add(q, int(rng.integers(g)), gates.u1(float(rng.uniform(0.1, 3))))
      elif r < 0.8:
        add(int((q + 1 + rng.integers(n - 1)) % n) if rng.random() < 0.5 else None, q, [gates.hadamard(), gates.tgate(), gates.vgate()][int(rng.integers(3))])
    for q in range(g):                       # dense on a shard qubit: exchange, with rank-dependent sweeps queued in front of it
      add(None, q, gates.yroot())
      add(int(rng.integers(g, n)), q, gates.vgate())
  o2, g2 = workloads.qft_stream(range(n)).arrays()
  return (np.concatenate([np.array(ops, dtype=np.int32), o2]),
          np.concatenate([np.array(gs).view(np.float64).reshape(-1, 8), g2]))


def _worker(rank, world, port, n, mode, out_dir, variant=False):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK='0', QCC_PRELOAD_TORCH='1')
  import torch  # noqa: F401  (first: one HIP runtime per process)
  import torch.distributed as dist
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from qcc_amd import sharded
  st = sharded.ShardedState(n, fusion=1, local_rank=0, chunk_amps=1 << 16, exchange=mode)
  assert type(st.eng).__module__ == 'qcc_amd.device'
  assert st.exchange_path == 'host-staged'          # the engine's own exchange, rounds carried by gloo
  assert st.relayout is True                        # the engines own their shards and re-lay them out (all ranks agreed)
  ops, g8 = _variant_stream(n, st.g, 11) if variant else _stream(n, 5)
  st.init_basis(0b101101)
  st.run_stream(ops, g8)
  st.flush()
  full = st.gather_logical()
  n2 = st.norm2_global()
  if rank == 0:
    xs = st.stats()
    np.savez(os.path.join(out_dir, 'res.npz'), psi=full, norm2=n2, exchanges=st.exchanges, rounds=xs['exchange_rounds'],
             overlapped=xs['sweeps_overlapped_with_exchange'], packed=st.eng.exchange_stats()['rounds_packed'])
  dist.barrier()
  st.close()
  dist.destroy_process_group()


@pytest.mark.parametrize('world,mode', [(2, 'alltoall'), (4, 'alltoall'), (4, 'pairwise')])
def test_sharded_hip_engines_one_gpu(oracle, tmp_path, world, mode):
  import torch.multiprocessing as mp
  n = 18
  port = _free_port()
  mp.spawn(_worker, args=(world, port, n, mode, str(tmp_path)), nprocs=world, join=True)
  res = np.load(tmp_path / 'res.npz')
  ops, g8 = _stream(n, 5)
  want = np.zeros(1 << n, dtype=np.complex128)
  want[0b101101] = 1
  oracle.run_stream(want, n, ops, g8)
  assert np.max(np.abs(res['psi'] - want)) < 1e-11
  assert abs(float(res['norm2']) - 1) < 1e-11
  assert int(res['exchanges']) >= 1 and int(res['rounds']) >= 1


@pytest.mark.parametrize('world,n', [(2, 22), (4, 21)])
def test_ranks_with_different_gate_lists_exchange_consistently(oracle, tmp_path, world, n):
  """Each rank drops the gates whose shard-bit control is 0 for it, so the ranks plan DIFFERENT op lists; the
  geometry of their sweeps, layouts and exchanges must still be identical (planner.h: ghosts), which the engine
  verifies at every exchange of the host-staged transport (engine.hip verify_geometry) -- and the amplitudes must
  be the oracle's.  Shards of 2^19 / 2^20 amplitudes: slabs, relayout sweeps and packed rounds all take part."""
  import torch.multiprocessing as mp
  port = _free_port()
  mp.spawn(_worker, args=(world, port, n, 'alltoall', str(tmp_path), True), nprocs=world, join=True)
  res = np.load(tmp_path / 'res.npz')
  g = world.bit_length() - 1
  ops, g8 = _variant_stream(n, g, 11)
  want = np.zeros(1 << n, dtype=np.complex128)
  want[0b101101] = 1
  oracle.run_stream(want, n, ops, g8)
  assert np.max(np.abs(res['psi'] - want)) < 1e-11
  assert abs(float(res['norm2']) - 1) < 1e-11
  assert int(res['exchanges']) >= 3 and int(res['rounds']) >= 3
