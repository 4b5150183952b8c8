"""The planner's output executed on the CPU (tests/plan_interp.py) against the oracle: random
circuits, every plan shape the engine can be switched to, single and sharded handles.  No GPU."""
import ctypes
import zlib

import numpy as np
import pytest

from qcc_amd import gates, native, workloads
from tests import plan_interp
from tests.oracle_lib import NO_CTL

_dp = ctypes.POINTER(ctypes.c_double)


def _rand_unitary(rng):
  m = rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))
  q, r = np.linalg.qr(m)
  return q * (np.diag(r) / np.abs(np.diag(r)))


def _stream(rng, n, ngates, gshard=0):
  v, yr, h = gates.vgate(), gates.yroot(), gates.hadamard()
  pool = [h, yr, v, np.conj(np.asarray(yr).reshape(2, 2).T), np.conj(np.asarray(v).reshape(2, 2).T),
          gates.tgate(), gates.sgate(), gates.pauli_z(), gates.u1(0.37), gates.rz(0.9), gates.pauli_x(),
          gates.ry(0.8), gates.rx(0.4), gates.pauli_y(), _rand_unitary(rng)]
  if rng.random() < 0.3:     # operators the reference also feeds through apply1 (projectors, ladder operators, scalings)
    pool += [np.array([[1, 0], [0, 0]]), np.array([[0, 0], [0, 1]]), np.array([[0, 1], [0, 0]]), np.array([[0, 0], [1, 0]]),
             np.array([[0.5, 0], [0, 2.0]]), np.array([[0, 0], [0, 0]]), np.array([[2.0, 0], [0, 2.0]]),
             np.array([[1, 1], [1, 1]]) * 0.5]
  out = []
  for _ in range(ngates):
    g = np.asarray(pool[int(rng.integers(len(pool)))], dtype=np.complex128).reshape(4)
    diag = g[1] == 0 and g[2] == 0
    t = int(rng.integers(n))
    if t < gshard and not diag:
      t = gshard + t % (n - gshard)
    ctl = []
    if rng.random() < 0.5:
      k = 1 + (int(rng.integers(1, 3)) if rng.random() < 0.25 else 0)
      others = [q for q in range(n) if q != t]
      ctl = [int(c) for c in rng.choice(others, size=k, replace=False)]
    out.append((ctl, t, g))
  return out


def _oracle_apply(oracle, psi, n, stream):
  idx = np.arange(1 << n, dtype=np.uint64)
  for ctl, t, g in stream:
    if not ctl:
      oracle.apply1(psi, g, n, t)
    elif len(ctl) == 1:
      oracle.applyc(psi, g, n, ctl[0], t)
    else:
      mask = np.ones(1 << n, dtype=bool)
      for c in ctl:
        mask &= ((idx >> np.uint64(n - 1 - c)) & np.uint64(1)).astype(bool)
      tmp = psi.copy()
      oracle.apply1(tmp, g, n, t)
      psi[mask] = tmp[mask]


def _planned(n, nloc, shard, stream, bw=128):
  lib = native.load()
  h = ctypes.c_void_p()
  native.check(lib.qh_create_dry(nloc, bw, ctypes.byref(h)))
  if nloc != n:
    native.check(lib.qh_set_shard(h, n, shard))
  native.check(lib.qh_set_fusion(h, native.QH_FUSE_SWEEP))
  for ctl, t, g in stream:
    cm = 0
    for c in ctl:
      cm |= 1 << (n - 1 - c)
    g8 = np.ascontiguousarray(g).view(np.float64)
    native.check(lib.qh_apply_bits(h, cm, n - 1 - t, g8.ctypes.data_as(_dp)))
  sweeps, _ = plan_interp.export_plan(h)
  lib.qh_destroy(h)
  return sweeps


ENVS = [{}, {'QH_WAVE_BITS': '2'}, {'QH_WAVE_BITS': '0'}, {'QH_LANE_VALU': '2'}, {'QH_LANE_VALU': '2', 'QH_WAVE_BITS': '2'},
        {'QH_DEFER_DIAG': '0', 'QH_BFLY': '0'}, {'QH_SWEEP_RB': '3'}, {'QH_SPLIT_LANES': '0'}, {'QH_PROPAGATE_X': '0'}, {'QH_ROT_FUSE': '0'},
        {'QH_SEATS': '2'}, {'QH_SEATS': '2', 'QH_LANE_VALU': '2', 'QH_WAVE_BITS': '2'}, {'QH_SEATS': '0'}]


@pytest.mark.parametrize('env', ENVS, ids=lambda e: ','.join(f'{k[3:]}={v}' for k, v in e.items()) or 'default')
def test_planned_sweeps_equal_the_oracle(oracle, monkeypatch, env):
  for k, v in env.items():
    monkeypatch.setenv(k, v)
  rng = np.random.default_rng(zlib.crc32(repr(sorted(env.items())).encode()))
  for case in range(14):
    n = int(rng.integers(10, 16))
    gshard = int(rng.integers(0, 3)) if case % 3 == 2 else 0
    stream = _stream(rng, n, int(rng.integers(20, 220)), gshard)
    psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi = (psi / np.linalg.norm(psi)).astype(np.complex128)
    want = psi.copy()
    _oracle_apply(oracle, want, n, stream)
    nloc = n - gshard
    got = np.empty_like(psi)
    for shard in range(1 << gshard):
      sweeps = _planned(n, nloc, shard, stream)
      part = psi[shard << nloc: (shard + 1) << nloc].copy()
      plan_interp.run_plan(part, sweeps, nloc, shard)
      got[shard << nloc: (shard + 1) << nloc] = part
    err = float(np.max(np.abs(got - want)))
    assert err < 1e-11, (env, case, n, gshard, err)


@pytest.mark.parametrize('env', [{}, {'QH_WAVE_BITS': '2'}, {'QH_LANE_VALU': '2'}], ids=['default', 'WAVE_BITS=2', 'LANE_VALU=2'])
def test_planned_sweeps_complex64_geometry(oracle, monkeypatch, env):
  """complex64 tiles have FOUR fixed low lane bits (16 x 8 B = one line) and two movable ones: the
  plan for a 64-bit-wide handle, executed in double precision, must equal the oracle too."""
  for k, v in env.items():
    monkeypatch.setenv(k, v)
  rng = np.random.default_rng(zlib.crc32(repr(sorted(env.items())).encode()) + 64)
  for case in range(10):
    n = int(rng.integers(10, 16))
    stream = _stream(rng, n, int(rng.integers(20, 220)))
    psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi = (psi / np.linalg.norm(psi)).astype(np.complex128)
    want = psi.copy()
    _oracle_apply(oracle, want, n, stream)
    got = psi.copy()
    sweeps = _planned(n, n, 0, stream, bw=64)
    assert all(sp['lane_low'] == 4 for sp in sweeps)
    plan_interp.run_plan(got, sweeps, n)
    assert float(np.max(np.abs(got - want))) < 1e-11, (env, case, n)


def test_reference_workload_plans_equal_the_oracle(oracle):
  """QFT, supremacy and one Grover iteration at sizes the CPU handles: plan -> NumPy == oracle."""
  for n, sb in ((14, workloads.qft_stream(range(14))), (14, workloads.supremacy_stream(14, 12, seed=3)),
                (14, workloads.grover_stream(7, [1, 0, 1, 1, 0, 0, 1], iterations=1))):
    ops, g8 = sb.arrays()
    stream = [([] if c == NO_CTL else [int(c)], int(t), g.view(np.complex128).copy()) for (c, t), g in zip(ops, g8)]
    rng = np.random.default_rng(5)
    psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi = (psi / np.linalg.norm(psi)).astype(np.complex128)
    want = psi.copy()
    oracle.run_stream(want, n, ops, g8)
    got = psi.copy()
    plan_interp.run_plan(got, _planned(n, n, 0, stream), n)
    assert np.max(np.abs(got - want)) < 1e-11


@pytest.mark.parametrize('bw', [128, 64])
@pytest.mark.parametrize('env', [{}, {'QH_LANE_VALU': '2'}, {'QH_WAVE_BITS': '2'}, {'QH_SEATS': '2'}],
                         ids=['default', 'LANE_VALU=2', 'WAVE_BITS=2', 'SEATS=2'])
def test_qft_phase_ladders_become_factor_trees(oracle, monkeypatch, env, bw):
  """A QFT's controlled-phase ladder between one register bit and the others is ONE group with bit factors
  (DG_BITFAC, planner.h fuse_bit_factors); the plan still computes the QFT (QFT and inverse on random states, both
  element widths' geometries, lane butterflies on either path, lane / wave exchanges left in place)."""
  for k, v in env.items():
    monkeypatch.setenv(k, v)
  rng = np.random.default_rng(20 + bw)
  for n in (13, 15):
    qops, qg = workloads.qft_stream(range(n)).arrays()
    stream = [([] if int(c) == NO_CTL else [int(c)], int(t), qg[k].view(np.complex128).copy()) for k, (c, t) in enumerate(qops)]
    if n == 15:   # ... followed by the inverse circuit
      stream = stream + [(c, t, np.conj(g.reshape(2, 2).T).reshape(4)) for c, t, g in reversed(stream)]
    psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi = (psi / np.linalg.norm(psi)).astype(np.complex128)
    want = psi.copy()
    _oracle_apply(oracle, want, n, stream)
    sweeps = _planned(n, n, 0, stream, bw=bw)
    nbf = sum(1 for sp in sweeps for g in sp['groups'] if int(g['flags']) & plan_interp.DG_BITFAC)
    assert nbf >= 3, nbf
    got = psi.copy()
    plan_interp.run_plan(got, sweeps, n, 0)
    assert float(np.max(np.abs(got - want))) < 1e-11


GEOMETRY_KEYS = ('rb', 'regpos', 'regpos_store', 'lanehi', 'nwave', 'wavepos', 'fixed_ones', 'ntiles', 'lane_low', 'relayout',
                 'dest_pos', 'seat', 'seat_store', 'seat_dest', 'wavepos_store', 'reg_dest', 'wave_dest', 'unit_runs', 'final_pos')


def _shard_variant_stream(rng, n, ngates, gshard):
  """Random circuit with plenty of RANK-DEPENDENT gates: dense gates under shard-bit controls (dropped on the
  ranks whose bit is 0), X gates under shard-bit controls, diagonal gates whose target is a shard bit."""
  base = _stream(rng, n, ngates, gshard)
  out = []
  for ctl, t, g in base:
    r = rng.random()
    if r < 0.25 and t >= gshard:                      # add a shard-bit control
      c = int(rng.integers(gshard))
      ctl = [c] + [q for q in ctl if q != c]
    elif r < 0.35 and t >= gshard:                    # X / H under ONLY a shard-bit control
      g = np.asarray(gates.pauli_x() if rng.random() < 0.5 else gates.hadamard(), dtype=np.complex128).reshape(4)
      ctl = [int(rng.integers(gshard))]
    elif r < 0.45:                                    # diagonal gate ON a shard bit, maybe under a local control
      g = np.asarray(gates.u1(float(rng.uniform(0.1, 3))) if rng.random() < 0.7 else gates.pauli_z(), dtype=np.complex128).reshape(4)
      t = int(rng.integers(gshard))
      ctl = [int(rng.integers(gshard, n))] if rng.random() < 0.6 else []
    out.append((ctl, t, g))
  return out


@pytest.mark.parametrize('env', [{}, {'QH_WAVE_BITS': '2'}, {'QH_LANE_VALU': '2'}, {'QH_RELAYOUT': '0'}],
                         ids=['default', 'WAVE_BITS=2', 'LANE_VALU=2', 'RELAYOUT=0'])
def test_every_rank_plans_the_same_geometry(oracle, monkeypatch, env):
  """The ranks of a sharded state exchange amplitudes block by block: they must agree on the sweeps, the tile
  bits of each, the slabs and -- with relayout sweeps -- on where every index bit lives afterwards, although
  each rank executes a different subset of the gates.  The planner keeps rank-dependent gates
  as ghosts (planner.h GateRec::ghost); here: identical geometry on every shard, and the right amplitudes."""
  for k, v in env.items():
    monkeypatch.setenv(k, v)
  rng = np.random.default_rng(zlib.crc32(repr(sorted(env.items())).encode()) + 7)
  for case in range(10):
    gshard = 1 + case % 3
    n = int(rng.integers(11, 15)) + gshard
    nloc = n - gshard
    stream = _shard_variant_stream(rng, n, int(rng.integers(30, 200)), gshard)
    psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi = (psi / np.linalg.norm(psi)).astype(np.complex128)
    want = psi.copy()
    _oracle_apply(oracle, want, n, stream)
    got = np.empty_like(psi)
    ref_geom = None
    for shard in range(1 << gshard):
      sweeps = _planned(n, nloc, shard, stream)
      geom = [[(k, sp[k]) for k in GEOMETRY_KEYS] for sp in sweeps]
      if ref_geom is None:
        ref_geom = geom
      assert geom == ref_geom, (env, case, shard, 'the ranks would disagree about an exchange')
      part = psi[shard << nloc: (shard + 1) << nloc].copy()
      plan_interp.run_plan(part, sweeps, nloc, shard)
      got[shard << nloc: (shard + 1) << nloc] = part
    err = float(np.max(np.abs(got - want)))
    assert err < 1e-11, (env, case, n, gshard, err)


def test_searched_plans_are_rank_invariant(oracle, monkeypatch):
  """plan_best picks among the candidates of its search by a predicted time that depends on the ops a rank really executes
  -- a gate that is a ghost on this rank has left the op list -- so a sharded handle must choose by something every rank sees
  alike (fewest sweeps, task order).  Layered circuits (the supremacy family: the search finds several tilings) over 18-20
  qubits on 4 ranks, with extra dense gates under shard-bit controls (live on some ranks, ghosts on others): identical
  geometry on every shard, right amplitudes."""
  monkeypatch.setenv('QH_PLAN_SEARCH_STEPS', '400000')
  monkeypatch.setenv('QH_PLAN_SEARCH_STREAMS', '6')
  rng = np.random.default_rng(2606)
  searched = 0
  for case, (n, seed) in enumerate(((18, 0), (19, 1), (20, 2), (18, 3))):
    gshard = 2
    nloc = n - gshard
    sops, sg = workloads.supremacy_stream(nloc, 20, seed=seed).arrays()     # the layered circuit on the LOCAL qubits gshard .. n-1
    stream = [([] if c == NO_CTL else [int(c) + gshard], int(t) + gshard, g.view(np.complex128).reshape(2, 2).copy()) for (c, t), g in zip(sops, sg)]
    # rank-dependent dense gates: random unitaries on local qubits under a control on a shard qubit (qubits 0, 1 are the shard bits)
    for k in range(48):
      q, _ = np.linalg.qr(rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)))
      stream.insert(int(rng.integers(30, len(stream))), ([int(rng.integers(0, gshard))], int(rng.integers(gshard, n)), q))
    psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi = (psi / np.linalg.norm(psi)).astype(np.complex128)
    want = psi.copy()
    _oracle_apply(oracle, want, n, stream)
    got = np.empty_like(psi)
    ref_geom = None
    monkeypatch.setenv('QH_PLAN_SEARCH', '0')
    greedy = len(_planned(n, nloc, 0, stream))
    monkeypatch.setenv('QH_PLAN_SEARCH', '1')
    for shard in range(1 << gshard):
      sweeps = _planned(n, nloc, shard, stream)
      geom = [[(k, sp[k]) for k in GEOMETRY_KEYS] for sp in sweeps]
      if ref_geom is None:
        ref_geom = geom
        searched += len(sweeps) < greedy
      assert geom == ref_geom, (case, shard, 'the ranks would disagree about an exchange')
      part = psi[shard << nloc: (shard + 1) << nloc].copy()
      plan_interp.run_plan(part, sweeps, nloc, shard)
      got[shard << nloc: (shard + 1) << nloc] = part
    assert float(np.max(np.abs(got - want))) < 1e-11, (case, n)
  assert searched >= 1, 'no case in which the search changed the plan: the test does not test what it is for'


def test_level_search_saves_sweeps_and_keeps_the_amplitudes(oracle, monkeypatch):
  """planner.h search_levels: for layered circuits the greedy tile choice is not the best one; a budgeted local search over the
  nested cuts of the circuit (deterministic, counted in label changes, a portfolio of host threads) looks for fewer sweeps.
  BASELINE config 3 (30 qubits, depth 20): seeds 0-3 run in 4 sweeps (greedy 5 6 6 6), seed 5 in 5 (greedy 8 with one wave bit, 7 with two; an integer
  program says that is its minimum, profiles/r06/level_search.txt); the plans it produces must of course still compute the
  circuit (here at 15-17 qubits, through NumPy)."""
  from tests.test_planner_cpu import _plan
  monkeypatch.setenv('QH_PLAN_SEARCH_STEPS', '2000000')      # (pinned: the default scales with the sweep time)
  monkeypatch.setenv('QH_PLAN_SEARCH_STREAMS', '6')          # (pinned: a host with few cores gets fewer by default)
  streams = {seed: workloads.supremacy_stream(30, 20, seed=seed).arrays() for seed in (0, 1, 2, 3, 5)}
  ops, g8 = streams[0]
  monkeypatch.setenv('QH_PLAN_SEARCH', '0')
  greedy = [len(_plan(30, *streams[seed])['sweeps']) for seed in (0, 1, 2, 3, 5)]
  monkeypatch.setenv('QH_PLAN_SEARCH', '1')
  searched = {seed: _plan(30, *streams[seed]) for seed in (0, 1, 2, 3, 5)}
  assert greedy == [5, 6, 6, 6, 8] and [len(searched[s]['sweeps']) for s in (0, 1, 2, 3, 5)] == [4, 4, 4, 4, 5]
  for seed, p in searched.items():
    assert sum(s['gates'] for s in p['sweeps']) + p['noop_gates'] == len(streams[seed][0])
  # twelve-bit tiles (one wave bit) only: seeds 0 and 1 have a 4-sweep tiling there too, seed 2 has none (five)
  monkeypatch.setenv('QH_PLAN_SEARCH_WB2', '0')
  monkeypatch.setenv('QH_PLAN_ONLY_WB', '1')
  assert [len(_plan(30, *streams[s])['sweeps']) for s in (1, 2)] == [4, 5]
  monkeypatch.delenv('QH_PLAN_SEARCH_WB2')
  monkeypatch.delenv('QH_PLAN_ONLY_WB')
  # deterministic: same circuit, same tiles -- with the streams on host threads and one after the other
  again = _plan(30, *streams[2])
  assert [s['regpos'] for s in again['sweeps']] == [s['regpos'] for s in searched[2]['sweeps']]
  monkeypatch.setenv('QH_PLAN_SEARCH_THREADS', '0')
  serial = _plan(30, *streams[2])
  assert [(s['regpos'], s['lanehi'], s['wavepos']) for s in serial['sweeps']] == [(s['regpos'], s['lanehi'], s['wavepos']) for s in again['sweeps']]
  monkeypatch.delenv('QH_PLAN_SEARCH_THREADS')
  monkeypatch.setenv('QH_PLAN_SEARCH_STEPS', '4000000')      # (small states get no budget by default: a sweep is cheap there)
  rng = np.random.default_rng(16)
  for n, seed in ((16, 0), (17, 3), (15, 1)):
    sops, sg = workloads.supremacy_stream(n, 20, seed=seed).arrays()
    stream = [([] if c == NO_CTL else [int(c)], int(t), g.view(np.complex128).copy()) for (c, t), g in zip(sops, sg)]
    psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi = (psi / np.linalg.norm(psi)).astype(np.complex128)
    want = psi.copy()
    oracle.run_stream(want, n, sops, sg)
    monkeypatch.setenv('QH_PLAN_SEARCH', '0')
    base = _planned(n, n, 0, stream)
    monkeypatch.setenv('QH_PLAN_SEARCH', '1')
    sweeps = _planned(n, n, 0, stream)
    assert len(sweeps) <= len(base)
    got = psi.copy()
    plan_interp.run_plan(got, sweeps, n)
    assert np.max(np.abs(got - want)) < 1e-11, (n, seed, len(base), len(sweeps))
