"""BASELINE.json config 5 on ONE MI355X: the 36-qubit QFT sharded over 8 GPUs, shard by shard.

A shard of that run is 2^33 amplitudes (128 GiB) behind a handle that knows it is shard r of a
36-qubit state (qh_set_shard).  One GPU can hold one such shard, so each piece of the 8-GPU job
is run here exactly as a rank would run it, and checked against the exact product-state oracle
(tests/product_oracle.py; the QFT of a basis state never entangles) and the closed form
(SURVEY 8c/8d config 5):

  local part   every gate before the first H on a shard qubit -- 630 - 3 CU1 and 33 H, controls
               and diagonal targets on shard bits resolved from the shard index (no traffic);
  exchange     shard bits <-> local bits is pure data movement (tested with RCCL/gloo elsewhere);
               here its RESULT is injected: the state after the local part is a product state, and
               qh_init_product builds the slice shard r holds after the exchange, in the bit map
               the exchange leaves (qh_remap_swap);
  final part   the last 3 H and 3 CU1, now on local bits of the 2^33-amplitude shard.
"""
import numpy as np
import pytest

from qcc_amd import device, native, workloads
from tests.product_oracle import ProductState

pytestmark = pytest.mark.gpu

N, G = 36, 3
NLOC = N - G


def _alloc(nloc):
  try:
    return device.DeviceState(nloc, 128, fusion=native.QH_FUSE_SWEEP)
  except native.QhError as e:
    if e.code == native.QH_ERR_NOMEM:
      pytest.skip(f'cannot allocate a 2^{nloc}-amplitude shard on this box: {e}')
    raise


def _split_stream():
  ops, g8 = workloads.qft_stream(range(N)).arrays()
  assert len(ops) == 666
  dense_on_shard = [k for k in range(len(ops)) if ops[k, 0] == workloads.NO_CTL and ops[k, 1] < G]
  assert len(dense_on_shard) == 3           # exactly 3 exchanging gates (H on qubits 2, 1, 0)
  cut = dense_on_shard[0]
  assert cut == 666 - 6
  return ops, g8, cut


def _windows(st, rng, count=12, width=2048):
  """(global physical index, amplitude) samples of the shard: a few contiguous windows."""
  nloc = st.nbits
  offs = [0, (1 << nloc) - width] + [int(o) for o in rng.integers(0, (1 << nloc) - width, size=count)]
  idx = np.concatenate([np.arange(o, o + width, dtype=np.uint64) for o in offs])
  amp = np.concatenate([st.download(o, width) for o in offs])
  return idx, amp


def _phys_to_logical(st, shard, local_idx, nglob):
  perm = (np.zeros(nglob, dtype=np.int32))
  import ctypes
  native.check(st.lib.qh_get_bitmap(st.h, perm.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
  phys = (np.uint64(shard) << np.uint64(st.nbits)) | local_idx
  out = np.zeros_like(phys)
  for b in range(nglob):
    out |= ((phys >> np.uint64(perm[b])) & np.uint64(1)) << np.uint64(b)
  return out


@pytest.mark.parametrize('shard', [0, 5, 7])
def test_config5_shard_local_part(shard):
  """Shard r of the 36-qubit register runs everything up to the exchange; the input basis state
  lives on this shard (top three bits of x = r), so the shard holds the whole non-trivial state."""
  ops, g8, cut = _split_stream()
  x = (shard << NLOC) | (0x1B2CB9A5E3 & ((1 << NLOC) - 1))
  with _alloc(NLOC) as st:
    st.set_shard(N, shard)
    st.init_basis(x)
    st.run_stream(ops[:cut], g8[:cut])
    st.flush()
    s = st.stats()
    assert s['gates_submitted'] == cut
    n2 = st.norm2()
    assert abs(n2 - 1.0) < 1e-9
    ps = ProductState(N, x)
    ps.run(ops, g8, 0, cut)
    idx, amp = _windows(st, np.random.default_rng(50 + shard))
    want = ps.amplitudes(_phys_to_logical(st, shard, idx, N))
    assert np.max(np.abs(want)) > 1e-6
    assert np.max(np.abs(amp - want)) < 1e-10
    # a shard the input does NOT live on stays empty through the local part, whatever its index
    other = (shard + 3) % 8
    st.set_shard(N, other)
    st.init_basis(x)
    st.run_stream(ops[:cut], g8[:cut])
    assert st.norm2() == 0.0


@pytest.mark.parametrize('shard', [0, 5, 7])
def test_config5_shard_after_exchange(shard):
  """The slice rank r holds after the all-to-all (shard bits <-> the top three local bits), built
  in place from the exact intermediate product state, then the final 3 H + 3 CU1 on 2^33 amplitudes;
  checked against the closed form of the 36-qubit QFT."""
  ops, g8, cut = _split_stream()
  x = 0xA2CB9A5E3 | (5 << NLOC)
  ps = ProductState(N, x)
  ps.run(ops, g8, 0, cut)
  with _alloc(NLOC) as st:
    st.set_shard(N, shard)
    for k in range(G):                       # what ShardedState._exchange_all records
      st.remap_swap(NLOC + k, NLOC - G + k)
    st.init_product(ps.factors())
    n2 = st.norm2()
    assert abs(n2 - 1.0 / 8) < 1e-10          # every rank holds 1/8 of the norm after the exchange
    idx, amp = _windows(st, np.random.default_rng(60 + shard), count=4)
    logical = _phys_to_logical(st, shard, idx, N)
    assert np.max(np.abs(amp - ps.amplitudes(logical))) < 1e-12
    st.run_stream(ops[cut:], g8[cut:])        # H(2) CU1(2,1) CU1(2,0) H(1) CU1(1,0) H(0): all local now
    st.flush()
    assert abs(st.norm2() - 1.0 / 8) < 1e-10
    idx, amp = _windows(st, np.random.default_rng(70 + shard))
    logical = _phys_to_logical(st, shard, idx, N)
    want = workloads.qft_analytic(N, x, logical)
    assert np.max(np.abs(amp - want)) < 1e-10


def test_config5_shard_size_through_sharded_layer():
  """2^33 amplitudes through qcc_amd.sharded.ShardedState with one rank (the N=1 point of the
  33/34/35/36-qubit ladder): same code path as the multi-GPU bench, RCCL process group of size 1."""
  import os
  import socket
  import torch
  import torch.distributed as dist
  from qcc_amd import sharded
  if torch.cuda.mem_get_info(0)[0] < (140 << 30):
    pytest.skip('not enough free HBM for a 128 GiB shard')
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
  n = NLOC
  st = sharded.ShardedState(n, fusion=1, local_rank=0)
  try:
    ops, g8 = workloads.qft_stream(range(n)).arrays()
    x = 0x12CB9A5E3 & ((1 << n) - 1)
    st.init_basis(x)
    st.run_stream(ops, g8)
    st.flush()
    assert abs(st.norm2_global() - 1.0) < 1e-9
    rng = np.random.default_rng(80)
    for i in [0, (1 << n) - 1] + [int(v) for v in rng.integers(0, 1 << n, size=24)]:
      a = st.amplitude_local(i)
      assert a is not None and abs(a - workloads.qft_analytic(n, x, [i])[0]) < 1e-10
  finally:
    st.close()
    dist.destroy_process_group()


def test_config5_shard_per_gate_kernels():
  """The unfused kernels on a 2^33-amplitude shard (2^32 pairs per launch: more work items than
  one HIP launch may have threads -- the grid is capped and the kernels stride): the whole local
  part of the 36-qubit QFT, one kernel per gate (about 10 s of HBM traffic)."""
  ops, g8, cut = _split_stream()
  x = (5 << NLOC) | (0x1B2CB9A5E3 & ((1 << NLOC) - 1))
  try:
    st = device.DeviceState(NLOC, 128, fusion=native.QH_FUSE_OFF)
  except native.QhError as e:
    if e.code == native.QH_ERR_NOMEM:
      pytest.skip(str(e))
    raise
  with st:
    st.set_shard(N, 5)
    st.init_basis(x)
    st.run_stream(ops[:cut], g8[:cut])
    ps = ProductState(N, x)
    ps.run(ops, g8, 0, cut)
    idx, amp = _windows(st, np.random.default_rng(90), count=6)
    want = ps.amplitudes(_phys_to_logical(st, 5, idx, N))
    assert np.max(np.abs(want)) > 1e-6
    assert np.max(np.abs(amp - want)) < 1e-10
    assert abs(st.norm2() - 1.0) < 1e-9
    s = st.stats()
    assert s['kernels_launched'] + s['gates_noop'] == cut
