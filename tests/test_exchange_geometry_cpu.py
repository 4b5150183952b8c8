"""BASELINE config 5 planned at its REAL geometry, without any device: the 666-gate 36-qubit QFT
(/root/reference/src/lib/circuit.py:320-328) on 8 shards of 2^33 amplitudes -- and the 34 / 35-qubit points of the
ladder on 2 / 4 shards -- routed by the very code a multi-GPU run uses (qcc_amd.sharded.ShardRouter) over planner-only
engine handles (qh_create_dry + qh_set_shard + qh_comm_init_dry).  Every rank decides for itself how its sweeps are
planned, which local bits leave in the exchange, how the exchange is cut into slabs and rounds, whether rounds are
packed, how much staging it needs; an RCCL exchange only works if all of them decide THE SAME.  No multi-GPU box has
ever been in reach of this build, so this is where that claim is checked at full size: identical geometry records and
signatures on all ranks, one exchange per QFT, 4-5 sweeps per QFT, staging within budget."""
import numpy as np
import pytest

from qcc_amd import sharded, workloads

GIB = 1 << 30


def _plan_ladder_point(n, world, reps=2, **kw):
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  per_rank = []
  for rank in range(world):
    sh = sharded.DryShard(n, world, rank, **kw)
    marks = []
    for _ in range(reps):                    # a loop over the circuit, as bench.py runs it: one flush per step
      sh.run_stream(ops, g8)
      sh.flush()
      s = sh.stats()
      marks.append((s['sweeps'], sh.exchanges))
    per_rank.append({'geometries': sh.geometries, 'marks': marks, 'perm': list(sh.perm), 'stats': sh.stats()})
    sh.close()
  return per_rank


@pytest.mark.parametrize('n,world', [(36, 8), (35, 4), (34, 2)])
def test_config5_every_rank_cuts_the_exchange_the_same_way(n, world):
  ranks = _plan_ladder_point(n, world)
  ref = ranks[0]
  nloc = n - int(np.log2(world))
  assert nloc == 33
  for r, rec in enumerate(ranks):
    assert rec['geometries'] == ref['geometries'], (r, 'exchange geometry differs from rank 0')
    assert rec['marks'] == ref['marks'] and rec['perm'] == ref['perm'], r
    assert rec['stats']['bytes_swept'] == ref['stats']['bytes_swept'], r
  geos = ref['geometries']
  (sweeps1, x1), (sweeps2, x2) = ref['marks']
  # one all-to-all per QFT: the g H gates on the shard qubits are the only gates that communicate
  assert x1 == 1 and x2 == 2, ref['marks']
  assert 4 <= sweeps1 <= 5 and 4 <= sweeps2 - sweeps1 <= 5, ref['marks']
  for geo in geos:
    assert geo['peers'] == world - 1
    assert geo['slabs'] == 8 and bin(geo['slab_mask']).count('1') == 3          # QH_EXCHANGE_SLAB_BITS default
    assert geo['last_sweep_split'] == 1 and geo['sweeps_before'] >= 3             # the sweep before the exchange overlaps it
    assert bin(geo['block_bits']).count('1') == int(np.log2(world))
    assert not (geo['block_bits'] & geo['slab_mask'])
    # 2^22 amplitudes per peer and round (64 MiB); two receive halves (+ two send halves when rounds are packed)
    assert geo['chunk_bits'] == 22
    assert geo['staging_bytes'] == (4 if geo['packed'] else 2) * (world - 1) * (16 << 22)
    assert geo['staging_bytes'] <= 1.75 * GIB
    # every amplitude that leaves is in exactly one round: slabs x rounds x peers x chunk = (P-1)/P of the shard
    assert geo['slabs'] * geo['rounds_per_slab'] * geo['peers'] << geo['chunk_bits'] == (world - 1) * (1 << nloc) // world
  # a different geometry (the second QFT starts from another layout) must give a different signature; same -> same
  sigs = [g['signature'] for g in geos]
  assert len(sigs) == 2
  if (geos[0]['slab_mask'], geos[0]['block_bits'], geos[0]['bitmap_after']) != (geos[1]['slab_mask'], geos[1]['block_bits'], geos[1]['bitmap_after']):
    assert sigs[0] != sigs[1]


def test_ranks_with_other_planner_switches_are_told_apart(monkeypatch):
  """What the signature is for: a rank whose planner runs with another switch plans other sweeps / layouts -- its
  signature differs, so verify_geometry stops the exchange before data moves (engine.hip)."""
  a = _plan_ladder_point(34, 2, reps=1)[0]['geometries'][0]
  monkeypatch.setenv('QH_EXCHANGE_SLAB_BITS', '2')
  b = _plan_ladder_point(34, 2, reps=1)[0]['geometries'][0]
  assert a['slabs'] == 8 and b['slabs'] == 4 and a['signature'] != b['signature']
  monkeypatch.delenv('QH_EXCHANGE_SLAB_BITS')
  monkeypatch.setenv('QH_ROT_FUSE', '0')            # same slabs and rounds, another planner switch: still told apart
  c = _plan_ladder_point(34, 2, reps=1)[0]['geometries'][0]
  assert c['signature'] != a['signature']


def test_slab_bits_are_capped_at_three(monkeypatch):
  """UnitPerm::slab_pos has three entries; QH_EXCHANGE_SLAB_BITS above that must not reach the kernel."""
  monkeypatch.setenv('QH_EXCHANGE_SLAB_BITS', '5')
  geo = _plan_ladder_point(34, 2, reps=1)[0]['geometries'][0]
  assert geo['slabs'] == 8 and bin(geo['slab_mask']).count('1') == 3


def test_strong_scaling_points_plan_consistently():
  """33 qubits on 2 / 4 / 8 ranks (the strong-scaling line of SURVEY 8d): smaller shards, same agreement."""
  for world in (2, 4, 8):
    ranks = _plan_ladder_point(33, world, reps=1)
    for rec in ranks[1:]:
      assert rec['geometries'] == ranks[0]['geometries'] and rec['perm'] == ranks[0]['perm']
    assert ranks[0]['marks'][0][1] == 1
