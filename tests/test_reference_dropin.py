"""Drop-in evidence, runnable only where the reference is mounted (this build
container): the reference's OWN unit tests and algorithms executed against
  (a) qcc_amd.lib installed as ``src.lib`` (the API mirror), and
  (b) the reference's unmodified src/lib with only ``libxgates`` replaced by
      qcc_amd/dropin/libxgates.py (the literal two-function boundary).
No GPU here, so gate execution is the oracle-backed stand-in (tests/fake_device.py);
the same runner accepts `gpu` / `dropin-gpu` on a machine that has both."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/src'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference not mounted')

REF_TESTS = ['circuit_test', 'state_test', 'tensor_test', 'helper_test', 'measure_test',
             'equalities_test', 'bell_test', 'ops_test']
# (algorithm, seconds it needs on the CPU stand-in are small for all of these)
ALGOS = ['arith_classic', 'arith_quantum', 'entanglement_swap', 'estimate_pi', 'hadamard_test',
         'hhl_2x2', 'inversion_test', 'pauli_rep', 'qram', 'quantum_mean', 'state_prep',
         'state_prep_mottonen', 'supremacy', 'counting', 'minimum_finding', 'teleportation',
         'superdense', 'swap_test', 'phase_estimation',
         # the rest of the qc-using algorithms
         'grover', 'order_finding', 'sat3', 'hhl', 'graph_coloring', 'vqe_simple', 'quantum_walk', 'quantum_pca',
         'hamiltonian_cycle']


def _run(path, mode, env=None):
  e = dict(os.environ, PYTHONPATH=ROOT, **(env or {}))
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'ref_runner.py'), path, mode],
                     capture_output=True, text=True, timeout=600, env=e)
  assert r.returncode == 0, (path, mode, r.stdout[-800:], r.stderr[-1500:])
  return r


@pytest.mark.parametrize('name', REF_TESTS)
def test_reference_unit_tests_on_api_mirror(name):
  r = _run(os.path.join(REF, 'lib', name + '.py'), 'cpu')
  assert 'OK' in r.stderr


@pytest.mark.parametrize('name', ['circuit_test', 'state_test'])
def test_reference_unit_tests_on_literal_libxgates_dropin(name):
  r = _run(os.path.join(REF, 'lib', name + '.py'), 'dropin-cpu')
  assert 'OK' in r.stderr


@pytest.mark.parametrize('name', ALGOS)
def test_reference_algorithms_on_api_mirror(name):
  _run(os.path.join(REF, name + '.py'), 'cpu')


@pytest.mark.parametrize('name', ['arith_quantum', 'supremacy', 'qram'])
def test_reference_algorithms_on_literal_libxgates_dropin(name):
  _run(os.path.join(REF, name + '.py'), 'dropin-cpu')


@pytest.mark.parametrize('name', ['circuit_test', 'measure_test', 'arith_quantum', 'supremacy', 'phase_estimation',
                                  'inversion_test', 'qram'])
def test_reference_code_through_the_planner(name):
  """The reference's own tests / algorithms with every flush planned by the engine's sweep planner
  (dry handle) and the plan executed with NumPy: the planner under the reference's real call patterns."""
  path = os.path.join(REF, 'lib', name + '.py') if name in REF_TESTS else os.path.join(REF, name + '.py')
  _run(path, 'plan', env={'QCC_TENSOR_WIDTH': '128'})


def test_complex128_width_on_api_mirror():
  _run(os.path.join(REF, 'lib', 'circuit_test.py'), 'cpu', env={'QCC_TENSOR_WIDTH': '128'})


@pytest.mark.parametrize('name', ['lib/state_test', 'lib/circuit_test', 'lib/measure_test', 'grover', 'counting', 'sat3',
                                  'minimum_finding', 'state_prep'])
def test_reference_code_with_the_state_mirror_on(name):
  """the reference's own State tests and its five direct callers of State.apply1 / applyc (grover.py:77,
  counting.py:54, sat3.py:129, minimum_finding.py:88, state_prep.py:53) with the opt-in device mirror of
  qcc_amd/lib/state.py switched ON for every State, however small (QCC_STATE_MIRROR_MIN_QUBITS=1): every look the
  reference's code takes at a State -- indexing, NumPy calls, prints, comparisons -- must see the gates."""
  r = _run(os.path.join(REF, name + '.py'), 'cpu', env={'QCC_TEST_STATE_MIRROR': '1', 'QCC_STATE_MIRROR_MIN_QUBITS': '1'})
  if name.startswith('lib/'):
    assert 'OK' in r.stderr
  import ast
  stats = ast.literal_eval(r.stderr.rsplit('state-mirror:', 1)[1].strip().splitlines()[0])
  if name not in ('lib/measure_test', 'lib/state_test'):   # (the other six drive a State directly: the mirror must have engaged)
    assert stats['gates'] > 0 and stats['uploads'] > 0 and stats['downloads'] > 0, stats
