"""TEST-ONLY: run a file of the reference (its tests or algorithms) with
``src.lib`` resolved to qcc_amd.lib.  Container-only (needs /root/reference).

usage: python tests/ref_runner.py <path/to/reference_file.py> [cpu|plan|gpu|dropin-cpu|dropin-gpu] [args...]
  cpu: qcc_amd.lib is `src.lib`; gates run on the oracle-backed stand-in (tests/fake_device.py)
  plan: as cpu, but every flush is planned by the engine's planner and the plan executed with NumPy
  gpu: qcc_amd.lib is `src.lib`; gates run on the MI355X
  dropin-*: the REFERENCE's own src/lib is used unmodified; only its `libxgates`
            import resolves to qcc_amd/dropin/libxgates.py (the literal boundary)
absl is not installed in this image: a flags/app/absltest stub is created in a
temp dir (flag values only, no arithmetic)."""
import os
import runpy
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def absl_stub():
  tmp = tempfile.mkdtemp(prefix='qcc_absl_')
  os.makedirs(os.path.join(tmp, 'absl', 'testing'))
  open(os.path.join(tmp, 'absl', '__init__.py'), 'w').close()
  open(os.path.join(tmp, 'absl', 'testing', '__init__.py'), 'w').close()
  with open(os.path.join(tmp, 'absl', 'flags.py'), 'w') as f:
    f.write('class _F:\n  pass\nFLAGS = _F()\n'
            'def _d(name, default, help=None, **kw):\n  setattr(FLAGS, name, default)\n'
            'DEFINE_integer = DEFINE_string = DEFINE_bool = DEFINE_boolean = DEFINE_float = _d\n')
  with open(os.path.join(tmp, 'absl', 'app.py'), 'w') as f:
    f.write('class UsageError(Exception):\n  pass\n'
            'def run(main):\n  import sys\n  main(sys.argv[:1])\n')
  with open(os.path.join(tmp, 'absl', 'testing', 'absltest.py'), 'w') as f:
    f.write('import unittest\n'
            'class TestCase(unittest.TestCase):\n'
            '  def assertLen(self, c, n):\n    self.assertEqual(len(c), n)\n'
            'def main():\n  unittest.main(argv=["x"])\n')
  return tmp


def main():
  target, mode = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'cpu')
  try:
    import absl  # noqa: F401
  except ImportError:
    sys.path.insert(0, absl_stub())
  if mode.startswith('dropin'):
    sys.path.insert(0, os.path.join(ROOT, 'qcc_amd', 'dropin'))
    sys.path.insert(0, '/root/reference')
    from qcc_amd.lib import backend
    if mode == 'dropin-cpu':
      from tests import fake_device
      backend.set_host_executor(fake_device.OracleHostExecutor())
    from absl import flags
    from src.lib import circuit  # the reference's own module
    import libxgates
    assert circuit.apply1 is libxgates.apply1, 'reference circuit.py did not bind our libxgates'
    if os.environ.get('QCC_TENSOR_WIDTH'):
      flags.FLAGS.tensor_width = int(os.environ['QCC_TENSOR_WIDTH'])
    sys.argv = [target] + sys.argv[3:]
    runpy.run_path(target, run_name='__main__')
    return
  import qcc_amd.lib as qlib
  qlib.install_as_src_lib()
  from qcc_amd.lib import backend, tensor
  if os.environ.get('QCC_TENSOR_WIDTH'):
    tensor.set_tensor_width(int(os.environ['QCC_TENSOR_WIDTH']))
  if mode in ('cpu', 'plan'):
    from tests import fake_device
    # plan: gates go through the engine's planner; the plan is executed with NumPy (no GPU)
    backend.set_device_factory(fake_device.PlanDevice if mode == 'plan' else fake_device.OracleDevice)
    backend.set_host_executor(fake_device.OracleHostExecutor())
  if os.environ.get('QCC_TEST_STATE_MIRROR') == '1':
    # the opt-in device mirror of directly driven States (qcc_amd/lib/state.py), on for every State of the run
    backend.set_state_mirror(True)
    import atexit
    from qcc_amd.lib import state as _st
    atexit.register(lambda: print('state-mirror:', _st.mirror_stats(), file=sys.stderr))
  # algorithms import helpers as `from src.lib import ...` (ours) and each other as
  # `from src import x`: expose the reference's src/ directory for the latter only.
  import src
  src.__path__.append(os.path.join('/root/reference', 'src'))
  sys.argv = [target] + sys.argv[3:]
  runpy.run_path(target, run_name='__main__')


if __name__ == '__main__':
  main()
