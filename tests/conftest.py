"""pytest configuration: markers + shared helpers (oracle loader, golden dir)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
  # One HIP runtime per process: the torch wheel bundles its own libamdhip64 and librccl, and the engine binds to
  # whichever HIP runtime is mapped first.  Collecting the whole suite imports torch (tests/test_sharded_gloo.py) before any
  # test loads the engine; a SUBSET that reaches torch only later (test_gpu_sharded spawning its workers, test_gpu_config5)
  # would leave the engine on the system runtime and the RCCL it finds mapped on torch's -- ncclCommInitRank then fails.
  # Same order for every selection: torch first.
  try:
    import torch  # noqa: F401
  except ImportError:
    pass


@pytest.fixture(scope='session')
def golden_dir():
  return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def oracle():
  from tests import oracle_lib
  return oracle_lib.load()
