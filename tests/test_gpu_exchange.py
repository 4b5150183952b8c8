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
@pytest.mark.parametrize('pack', ['auto', 'packed', 'direct'])
@pytest.mark.parametrize('n,bit,chunk,bw', [(22, 21, 1 << 14, 128), (22, 13, 1 << 10, 128), (20, 19, 0, 64),
                                            (24, 9, 1 << 16, 128)])
def test_loopback_exchange_equals_x_gate(oracle, monkeypatch, transport, pack, n, bit, chunk, bw):
  """`bit` is a LOGICAL bit: the handle owns its memory and keeps re-laying the state out while the communicator
  exists (one rank: nobody to disagree with), so the blocks' bit may sit anywhere when the exchange starts --
  rounds are then packed by a gather kernel; 'packed' / 'direct' force either way of moving a round."""
  if pack != 'auto':
    monkeypatch.setenv('QH_EXCHANGE_PACK', '1' if pack == 'packed' else '0')
  if pack == 'direct' and (n, bit) in ((22, 21), (24, 9)):
    # Direct rounds are what the engine chooses when the blocks' bits give long runs: forced on a re-laid-out state they
    # degenerate into 128-byte pieces (2^18 sends per exchange: 166 + 113 s of the round-5 suite for these two cases).  Here
    # they run on the layout they are made for (no relayout sweeps); the scrambled-layout form of the forced direct path stays
    # covered by the (22, 13, 2^10) and the complex64 cases.
    monkeypatch.setenv('QH_RELAYOUT', '0')
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
    st.flush()                         # (the owning handle has re-laid the state out by now)
    if transport == 'rccl':
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
  if pack == 'packed':
    assert xs['rounds_packed'] == xs['rounds']
  if pack == 'direct':
    assert xs['rounds_packed'] == 0
  if n >= 22:
    assert xs['sweeps_overlapped'] >= 1   # at least one neighbouring sweep was cut into slabs


@pytest.mark.parametrize('relayout,pack', [('1', 'auto'), ('1', 'packed'), ('0', 'auto')])
def test_loopback_exchange_at_shard_size(monkeypatch, relayout, pack, capsys):
  """Config 5's exchange leg at FULL shard size on one GPU (VERDICT r2, next #1a): a 2^33-amplitude handle with a
  1-rank RCCL communicator sends the two 64-GiB halves selected by its top logical bit to itself -- default
  chunk, 8 slabs, 2 x 256 rounds of grouped ncclSend/ncclRecv, the staging halves, the landing copies (or, with
  relayout sweeps, the gather / scatter kernels) and the event chain between the four streams -- with a queued
  33-qubit sweep in front of it (cut into slabs) and one behind it (started slab by slab).  The data movement is
  an X gate on that bit; the circuit around it is a QFT, so the exact product-state oracle checks the result."""
  from tests.product_oracle import ProductState
  monkeypatch.setenv('QH_RELAYOUT', relayout)
  if pack == 'packed':
    monkeypatch.setenv('QH_EXCHANGE_PACK', '1')
  n, bit = 33, 32
  try:
    st = device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP)
  except native.QhError as e:
    if e.code == native.QH_ERR_NOMEM:
      pytest.skip(f'cannot allocate 2^{n} amplitudes on this box: {e}')
    raise
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  cut = int(np.flatnonzero((ops[:, 0] == NO_CTL) & (ops[:, 1] == n - 1 - 21))[0])   # the H on logical bit 21: two sweeps' worth before it
  x = 0x12CB9A5E3 & ((1 << n) - 1)
  x_ops, x_g = _x_on_bit(n, bit)
  ps = ProductState(n, x)
  ps.run(ops, g8, 0, cut)
  ps.run(x_ops, x_g)
  ps.run(ops, g8, cut, len(ops))
  with st:
    st.comm_init(1, 0, device.DeviceState.comm_unique_id())
    st.init_basis(x)
    st.run_stream(ops[:cut], g8[:cut])     # queued: the exchange plans them and cuts the last sweep into slabs
    st.exchange_loopback(bit, 0)           # default chunk (2^22 amplitudes per peer and round)
    st.run_stream(ops[cut:], g8[cut:])     # the first sweep of these starts slab by slab as the slabs land
    st.flush()
    st.sync()
    xs = st.exchange_stats()
    s = st.stats()
    n2 = st.norm2()
    rng = np.random.default_rng(33)
    idx = np.concatenate([np.arange(o, o + 512, dtype=np.uint64) for o in
                          [0, (1 << n) - 512] + [int(v) for v in rng.integers(0, (1 << n) - 512, size=24)]])
    amp = np.array([st.amplitude(int(i)) for i in idx[::16]])      # by logical index: no re-layout pass
    with capsys.disabled():
      print(f'\n[exchange @2^33, QH_RELAYOUT={relayout} pack={pack}] span_ms={xs["span_ms"]:.1f} rounds={xs["rounds"]} packed={xs["rounds_packed"]} '
            f'slabs={xs["slabs"]} sweeps_overlapped={xs["sweeps_overlapped"]} sweeps={s["sweeps"]} '
            f'GB/s(one way)={xs["bytes_sent"] / max(xs["span_ms"], 1e-9) / 1e6:.0f}')
  want = ps.amplitudes(idx[::16])
  assert np.max(np.abs(want)) > 1e-6
  assert np.max(np.abs(amp - want)) < 1e-10
  assert abs(n2 - 1) < 1e-9
  assert xs['exchanges'] == 1 and xs['bytes_sent'] == (1 << n) * 16 and xs['slabs'] == 8
  assert xs['rounds'] == (1 << (n - 1)) >> 22            # half the shard per move, 2^22 amplitudes per round
  assert xs['sweeps_overlapped'] == 2                    # the sweep before and the sweep after ran slab by slab
  if pack == 'packed':
    assert xs['rounds_packed'] == xs['rounds']


def test_exchange_needs_a_communicator():
  with device.DeviceState(12, 128, fusion=native.QH_FUSE_SWEEP) as st:
    with pytest.raises(native.QhError):
      st.exchange_alltoall(9)
    st.comm_init_custom(1, 0, _self_round)
    with pytest.raises(native.QhError):
      st.exchange_loopback(12)           # not a local bit
    assert st.exchange_stats()['exchanges'] == 0


def test_watched_waits_poll_and_time_out(monkeypatch, oracle):
  """The watchdog of host waits (engine.hip wait_stream / poll_until_done): handles whose communicator has several RCCL
  ranks poll their stream instead of blocking, and give up with QH_ERR_COMM after QH_COMM_TIMEOUT_MS -- a missing peer
  or a mismatched round is an error message, not a hung job (VERDICT r3 #1a).  No second GPU here, so the one-rank
  loop-back is watched too (QH_COMM_WATCH_ALL=1): (1) the polling path returns the right amplitudes and geometry
  record, (2) with the timeout at zero a wait behind 28-qubit sweeps gives up at once, the handle refuses further work
  until it is re-initialised."""
  monkeypatch.setenv('QH_COMM_WATCH_ALL', '1')
  n, bit = 20, 17
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  x_ops, x_g = _x_on_bit(n, bit)
  want = np.zeros(1 << n, dtype=np.complex128)
  want[9] = 1
  for o, g in ((ops, g8), (x_ops, x_g), (ops, g8)):
    oracle.run_stream(want, n, o, g)
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.comm_init(1, 0, device.DeviceState.comm_unique_id())
    st.init_basis(9)
    st.run_stream(ops, g8)
    st.exchange_loopback(bit, 1 << 12)
    st.run_stream(ops, g8)
    st.sync()                                   # polled
    geo = st.exchange_geometry()
    assert geo['signature'] != 0 and geo['slabs'] * geo['rounds_per_slab'] * geo['peers'] << geo['chunk_bits'] == 1 << n
    assert geo['staging_bytes'] == (4 if geo['packed'] else 2) * geo['peers'] * (16 << geo['chunk_bits'])
    assert np.max(np.abs(st.download() - want)) < 1e-11
  n = 28
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.comm_init(1, 0, device.DeviceState.comm_unique_id())
    st.init_basis(3)
    st.sync()
    monkeypatch.setenv('QH_COMM_TIMEOUT_MS', '0')
    for _ in range(8):
      st.run_stream(ops, g8)
      st.flush()
    with pytest.raises(native.QhError) as ei:   # ~35 ms of sweeps are queued: the watched wait gives up first
      st.sync()
    assert ei.value.code == native.QH_ERR_COMM and 'still waiting' in str(ei.value)
    monkeypatch.setenv('QH_COMM_TIMEOUT_MS', '60000')
    with pytest.raises(native.QhError):         # poisoned: refuses work ...
      st.run_stream(ops, g8)
      st.flush()
    st.init_basis(3)                            # ... until re-initialised
    st.run_stream(ops, g8)
    st.sync()
    assert abs(st.norm2() - 1) < 1e-10
