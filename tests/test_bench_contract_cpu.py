"""bench.py's contract pieces that need no GPU: the workload each --gpus N maps to (BASELINE.json
configs 2 and 5) and the units of `value`."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
  spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_gpus_to_workload_ladder():
  b = _bench()
  assert b.default_qubits(1) == 30                       # config 2: 30-qubit QFT on one GPU (the headline)
  assert [b.default_qubits(n) for n in (2, 4, 8)] == [34, 35, 36]   # config 5: 128 GiB per GPU
  for n in (2, 4, 8):
    g = n.bit_length() - 1
    assert (16 << (b.default_qubits(n) - g)) == 128 << 30
  with pytest.raises(AssertionError):
    b.default_qubits(3)


def test_qft_gate_counts_of_the_ladder():
  from qcc_amd import workloads
  for n, gates in ((30, 465), (33, 561), (36, 666)):
    ops, _ = workloads.qft_stream(range(n)).arrays()
    assert len(ops) == gates == n * (n + 1) // 2


def test_memory_plan_and_step_prediction_of_the_ladder():
  """what a rank of BASELINE config 5 asks for, and what a step should cost (DESIGN 8's table is this
  function): 36 qubits on 8 GPUs = 128 GiB shard + 128 GiB second buffer + 1.75 GiB staging; one GPU = 3 sweeps, several
  = 4 sweeps + one all-to-all whose link time (shard / P per link) the slabs hide in part."""
  from qcc_amd import sharded
  m = sharded.memory_plan(36, 8)
  assert m['local_qubits'] == 33 and m['shard_bytes'] == 128 << 30 and m['second_buffer_bytes'] == 128 << 30
  assert m['staging_bytes'] == 4 * 7 * (1 << 22) * 16 == int(1.75 * 2 ** 30)
  assert m['need_relayout_bytes'] == 2 * (128 << 30) + m['staging_bytes'] < 288e9 and m['need_in_place_bytes'] == (128 << 30) + m['staging_bytes']
  assert sharded.memory_plan(30, 1)['staging_bytes'] == 0
  p1 = sharded.predict_step_ms(33, 1)
  assert p1['sweeps'] == 3 and p1['exchanges'] == 0 and abs(p1['expected_ms'] - 3 * 46.7) < 1e-9 and p1['link_ms'] == 0
  prev = None
  for n, w in ((34, 2), (35, 4), (36, 8)):
    p = sharded.predict_step_ms(n, w)
    assert p['sweeps'] == 4 and p['exchanges'] == 1 and abs(p['sweep_ms'] - 46.7) < 1e-9
    assert abs(p['link_ms'] - (128 << 30) / w / (153e9 * 0.7) * 1e3) < 1e-6
    assert p['best_case_ms'] <= p['expected_ms'] and p['expected_ms'] >= 4 * 46.7 + p['pack_ms'] - 1e-9
    assert prev is None or p['expected_ms'] < prev          # more peers = more links in parallel: the per-link share shrinks
    prev = p['expected_ms']
  # strong scaling: the shard halves per doubling, so do sweep and link time
  assert abs(sharded.predict_step_ms(33, 8)['sweep_ms'] - 46.7 / 8) < 1e-9
  # measured counts override the defaults (bench.py passes what the run had)
  assert sharded.predict_step_ms(36, 8, sweeps=5, exchanges=2)['expected_ms'] > sharded.predict_step_ms(36, 8)['expected_ms']


def test_memory_plan_check_refuses_before_allocating(monkeypatch):
  """A shard that does not fit the free device memory is refused BEFORE anything is allocated, with the plan in the
  message (bench.py turns it into the JSON "error" line); one that fits only without the second buffer says so."""
  import types
  from qcc_amd import sharded
  fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(is_available=lambda: True,
                                                                  mem_get_info=lambda dev: (200 << 30, 268 << 30)))
  me = types.SimpleNamespace(torch=fake_torch, dist=None, bit_width=128)
  plan = sharded.ShardedState._check_memory_plan(me, 33, 1, 0, 0, 1 << 22, True)        # 128 GiB shard: fits in place only
  assert plan['fits_in_place'] is True and plan['fits_relayout'] is False and plan['free_bytes'] == 200 << 30
  with pytest.raises(sharded.MemoryPlanError, match='exceed the 200.00 GiB free'):
    sharded.ShardedState._check_memory_plan(me, 34, 1, 0, 0, 1 << 22, True)             # 256 GiB shard
  plan = sharded.ShardedState._check_memory_plan(me, 34, 1, 0, 0, 1 << 22, False)       # stand-in engines: nothing to check
  assert plan['fits_in_place'] is None and plan['free_bytes'] is None
