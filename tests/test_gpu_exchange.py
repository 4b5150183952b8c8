"""The exchange step behind the C-ABI (qh_comm_* / qh_exchange_*) on ONE MI355X.

RCCL needs one GPU per rank, so the multi-rank data path cannot run here; what can run is every
piece of it on a 1-rank communicator: dlopen of RCCL, ncclCommInitRank, grouped ncclSend/ncclRecv
on the exchange stream, the two staging halves and their landing copies, the slab-wise launch of
the sweep before and the sweep after, and the HIP-event ordering between the three streams --
through qh_exchange_loopback, which sends the two halves of the shard selected by one local bit
to this rank itself and lands them exchanged.  That is exactly an X gate on that bit, so the
oracle can check the result amplitude by amplitude.  (World sizes 2 and 4 run through the
host-staged transport in tests/test_gpu_sharded.py.)"""
import numpy as np
import pytest

from qcc_amd import device, gates, native, workloads
from tests.oracle_lib import NO_CTL

pytestmark = pytest.mark.gpu


def _circuit(n, seed, count):
  rng = np.random.default_rng(seed)
  pool = [gates.hadamard(), gates.tgate(), gates.u1(0.37), gates.vgate(), gates.ry(0.3), gates.yroot(), gates.pauli_x()]
  ops, gs = [], []
  for _ in range(count):
    t = int(rng.integers(n))
    g = pool[int(rng.integers(len(pool)))]
    ops.append((int((t + 1 + rng.integers(n - 1)) % n), t) if rng.random() < 0.4 else (NO_CTL, t))
    gs.append(np.asarray(g, dtype=np.complex128).reshape(4))
  return np.array(ops, dtype=np.int32), np.array(gs).view(np.float64).reshape(-1, 8)


def _x_on_bit(n, bit):
  return (np.array([[NO_CTL, n - 1 - bit]], dtype=np.int32),
          np.asarray(gates.pauli_x(), dtype=np.complex128).reshape(1, 4).view(np.float64).reshape(1, 8))


def _self_round(peers, send, recv):
  assert all(p == 0 for p in peers)
  # two sends to self, two receives from self: matched in order
  for s, r in zip(send, recv):
    r[:] = s


@pytest.mark.parametrize('transport', ['rccl', 'host'])
@pytest.mark.parametrize('n,bit,chunk,bw', [(22, 21, 1 << 14, 128), (22, 13, 1 << 10, 128), (20, 19, 0, 64),
                                            (24, 9, 1 << 16, 128)])
def test_loopback_exchange_equals_x_gate(oracle, transport, n, bit, chunk, bw):
  a_ops, a_g = _circuit(n, 100 + bit, 40)
  b_ops, b_g = _circuit(n, 200 + bit, 40)
  q_ops, q_g = workloads.qft_stream(range(n)).arrays()
  want = np.zeros(1 << n, dtype=np.complex128)     # (complex64 runs are compared with the double-precision oracle)
  want[5] = 1
  x_ops, x_g = _x_on_bit(n, bit)
  for o, g in ((a_ops, a_g), (q_ops, q_g), (x_ops, x_g), (b_ops, b_g), (x_ops, x_g), (x_ops, x_g)):
    oracle.run_stream(want, n, o, g)
  with device.DeviceState(n, bw, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(5)
    st.run_stream(a_ops, a_g)
    st.flush()                         # (an owning handle may re-lay the state out here: the communicator
    if transport == 'rccl':            #  attaches to a canonical layout again)
      st.comm_init(1, 0, device.DeviceState.comm_unique_id())
    else:
      st.comm_init_custom(1, 0, _self_round)
    st.run_stream(q_ops, q_g)          # left queued: the exchange runs it, last sweep slab by slab
    st.exchange_loopback(bit, chunk)
    st.run_stream(b_ops, b_g)          # first sweep starts slab by slab as the slabs arrive
    st.flush()
    st.exchange_loopback(bit, chunk)   # nothing queued in front of this one
    st.exchange_loopback(bit, chunk)   # ... and back to back
    got = st.download()
    xs = st.exchange_stats()
    n2 = st.norm2()
  tol = 1e-11 if bw == 128 else 3e-5
  assert np.max(np.abs(got - want)) < tol
  assert abs(n2 - 1) < (1e-11 if bw == 128 else 1e-4)
  assert xs['exchanges'] == 3 and xs['rounds'] >= 3 and xs['bytes_sent'] == 3 * (1 << n) * (16 if bw == 128 else 8)
  assert xs['slabs'] >= 3 and xs['span_ms'] > 0
  if n >= 22:
    assert xs['sweeps_overlapped'] >= 1   # at least one neighbouring sweep was cut into slabs


def test_exchange_needs_a_communicator():
  with device.DeviceState(12, 128, fusion=native.QH_FUSE_SWEEP) as st:
    with pytest.raises(native.QhError):
      st.exchange_alltoall(9)
    st.comm_init_custom(1, 0, _self_round)
    with pytest.raises(native.QhError):
      st.exchange_loopback(12)           # not a local bit
    assert st.exchange_stats()['exchanges'] == 0
