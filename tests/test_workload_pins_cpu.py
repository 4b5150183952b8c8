"""Pins of the workload generators (qcc_amd/workloads.py) for BASELINE configs 3 and 4 against the native-call
traces recorded from the reference itself (tests/golden/g5_*.npz, g9_*.npz; tools/make_golden*.py ran
/root/reference/src/supremacy.py:123-158,208-253 and /root/reference/src/grover.py:124-168 and wrote down every
apply1 / applyc call with its gate doubles).  The bench and the full-size GPU tests replay these generators at 30 /
34 qubits, where no reference can run: what keeps them honest is that at the sizes the reference DID run the
generated stream is the recorded one -- same calls in the same order, same matrices.

Provenance note (also in DESIGN.md section 5): the g5 / g9 traces were recorded with the reference's Python package
imported under a three-module `absl` flag-holder stub (flag VALUES only; absl-py is not installed and cannot be);
the native calls themselves went through the reference's unmodified libxgates build."""
import os

import numpy as np
import pytest

from qcc_amd import workloads


def _load(golden_dir, name):
  return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize('name', ['g5_supremacy_n12_s0.npz', 'g5_supremacy_n14_s1.npz', 'g5_supremacy_n16_s2.npz',
                                  'g9_supremacy_n20_s0.npz'])
def test_supremacy_stream_is_the_recorded_reference_trace(golden_dir, name):
  g = _load(golden_dir, name)
  n, depth, seed = int(g['nbits']), int(g['depth']), int(g['seed'])
  ops, g8 = workloads.supremacy_stream(n, depth, seed=seed).arrays()
  assert ops.shape == g['ops'].shape, (name, ops.shape, g['ops'].shape)
  assert np.array_equal(ops, g['ops'])                 # same calls, same order, same qubits
  assert np.array_equal(g8, g['gates'])                # bit-identical gate doubles (H, T, V, Yroot, Z)
  # the generator must not depend on state left behind by an earlier call (random.seed is set inside)
  ops2, g82 = workloads.supremacy_stream(n, depth, seed=seed).arrays()
  assert np.array_equal(ops, ops2) and np.array_equal(g8, g82)


@pytest.mark.parametrize('name', ['g5_grover6.npz', 'g5_grover7.npz', 'g5_grover8.npz'])
def test_grover_stream_is_the_recorded_reference_trace(golden_dir, name):
  g = _load(golden_dir, name)
  nbits = int(g['nbits']) // 2
  marked = [int(b) for b in g['marked']]
  assert len(marked) == nbits
  ops, g8 = workloads.grover_stream(nbits, marked).arrays()          # default iterations = the reference's floor(pi/4 sqrt N)
  assert ops.shape == g['ops'].shape, (name, ops.shape, g['ops'].shape)
  assert np.array_equal(ops, g['ops'])
  # X, H, Z are exact; sqrt(X) and its adjoint come from scipy.linalg.sqrtm in the reference (circuit.py:238) and from a
  # closed form here: equal to rounding
  assert float(np.max(np.abs(g8 - g['gates']))) <= 1e-15
  # the recorded initial state is the basis state the generator names
  init = np.zeros(1 << (2 * nbits), dtype=np.complex128)
  init[workloads.grover_initial_index(nbits)] = 1
  assert np.array_equal(init, g['init'])


def test_supremacy_30_and_grover_34_stream_shapes():
  """The full-size streams the bench replays (configs 3 and 4): gate counts by class as SURVEY 8(a) A10 / A11 probed
  them on the reference (30 H; 80 T; 117 V / Yroot; 115 CZ for seed 0 -- one Grover iteration at nbits = 17: 322
  controlled + 84 single-qubit calls + the 18 initial H)."""
  ops, g8 = workloads.supremacy_stream(30, 20, seed=0).arrays()
  ctl = ops[:, 0] != workloads.NO_CTL
  diag = (g8[:, 2:6] == 0).all(axis=1)
  assert len(ops) == 342 and int(ctl.sum()) == 115 and int((~ctl & diag).sum()) == 80 and int((~ctl & ~diag).sum()) == 147
  nb = 17
  ops, g8 = workloads.grover_stream(nb, [1, 0] * 8 + [1], iterations=1).arrays()
  ctl = ops[:, 0] != workloads.NO_CTL
  assert int(ctl.sum()) == 322 and len(ops) - int(ctl.sum()) == 84 + (nb + 1)
