"""The product-state tracker (tests/product_oracle.py) against the C oracle: it is the analytic
oracle of the 33..36-qubit tests, so it is pinned here at sizes the oracle runs."""
import numpy as np
import pytest

from qcc_amd import workloads
from tests.product_oracle import ProductState


@pytest.mark.parametrize('n,x', [(5, 0b10110), (9, 0b101100101), (12, 0b101100101110)])
def test_tracker_matches_oracle_on_qft_prefixes(oracle, n, x):
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  for cut in (0, 1, n, len(ops) // 2, len(ops) - 3, len(ops)):
    want = np.zeros(1 << n, dtype=np.complex128)
    want[x] = 1
    oracle.run_stream(want, n, ops[:cut], g8[:cut])
    ps = ProductState(n, x)
    ps.run(ops, g8, 0, cut)
    got = ps.amplitudes(np.arange(1 << n))
    assert np.max(np.abs(got - want)) < 1e-14
  # full QFT: the closed form too
  assert np.max(np.abs(got - workloads.qft_analytic(n, x, np.arange(1 << n)))) < 1e-13


def test_tracker_refuses_entangling_gates():
  ps = ProductState(3, 0)
  h = np.array([1, 1, 1, -1]) / np.sqrt(2)
  ps.apply1(h, 0)
  ps.apply1(h, 1)
  with pytest.raises(ValueError):
    ps.applyc(np.array([0, 1, 1, 0]), 0, 1)
