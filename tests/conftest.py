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


# ---- the GPU selection (-m gpu) on a FRESH box ---------------------------------------------------------------------------
# The first RCCL communicator of a box pages librccl's code objects in from the image: 6 s on a warm box, 216 s on a cold one
# (round 6, gpurun_out/r06d: test_bench_sharded_world_of_one) -- a third of the suite's time limit for nothing.  So (1) the
# files whose tests create communicators run LAST, and (2) a helper process creates one 1-rank communicator in the background
# from the start of the session: by the time those tests run, the pages are in the cache.  No test's content changes.
_RCCL_LAST = ('test_gpu_bench_contract.py', 'test_gpu_config5.py', 'test_gpu_exchange.py', 'test_gpu_sharded.py')
_warmup = None


def _gpu_selected(config):
  expr = config.getoption('markexpr', default='') or ''
  return 'gpu' in expr and 'not gpu' not in expr


def pytest_collection_modifyitems(config, items):
  if not _gpu_selected(config):
    return
  name = lambda it: os.path.basename(str(it.fspath))   # noqa: E731
  # (the full-size file -- its 256-GiB Grover state leaves 12 GiB of the GPU free -- after the small ones: the helper's own
  #  context is gone or idle by then)
  first = [it for it in items if name(it) not in _RCCL_LAST and name(it) != 'test_gpu_fullsize.py']
  big = [it for it in items if name(it) == 'test_gpu_fullsize.py']
  last = [it for it in items if name(it) in _RCCL_LAST]
  items[:] = first + big + last


def pytest_sessionstart(session):
  global _warmup
  if not _gpu_selected(session.config) or not os.path.exists('/dev/kfd'):
    return
  import subprocess
  code = ('import torch\n'
          'from qcc_amd import device, native\n'
          'st = device.DeviceState(12, 128, fusion=native.QH_FUSE_SWEEP)\n'
          'st.comm_init(1, 0, device.DeviceState.comm_unique_id())\n'
          'st.close()\n')
  try:
    _warmup = subprocess.Popen([sys.executable, '-c', code], cwd=ROOT, env=dict(os.environ, QCC_PRELOAD_TORCH='1', PYTHONPATH=ROOT),
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
  except OSError:
    _warmup = None


def pytest_sessionfinish(session, exitstatus):
  if _warmup is not None and _warmup.poll() is None:
    _warmup.kill()


@pytest.fixture(scope='session')
def golden_dir():
  return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def oracle():
  from tests import oracle_lib
  return oracle_lib.load()
