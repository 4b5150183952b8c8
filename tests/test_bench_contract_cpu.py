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
