"""CPU tests of the sweep planner (qcc_amd/csrc/planner.h) through the C-ABI's
dry handle + qh_plan_json: no GPU needed to check what would be launched."""
import ctypes
import json

import numpy as np
import pytest

from qcc_amd import gates, native, workloads

_dp = ctypes.POINTER(ctypes.c_double)


def _plan(n, ops, g8, shard=None):
  lib = native.load()
  h = ctypes.c_void_p()
  native.check(lib.qh_create_dry(n if shard is None else shard[0], 128, ctypes.byref(h)))
  if shard is not None:
    native.check(lib.qh_set_shard(h, n, shard[1]))
  native.check(lib.qh_set_fusion(h, native.QH_FUSE_SWEEP))
  g8 = np.ascontiguousarray(g8, dtype=np.float64)
  for k in range(len(ops)):
    gp = ctypes.cast(g8.ctypes.data + 64 * k, _dp)
    c, t = int(ops[k, 0]), int(ops[k, 1])
    native.check(lib.qh_apply1(h, t, gp) if c == workloads.NO_CTL else lib.qh_applyc(h, c, t, gp))
  need = ctypes.c_uint64()
  lib.qh_plan_json(h, None, 0, ctypes.byref(need))
  buf = ctypes.create_string_buffer(need.value)
  lib.qh_plan_json(h, buf, need.value, None)
  lib.qh_destroy(h)
  return json.loads(buf.value.decode())


def test_qft30_is_three_sweeps_and_accounts_every_gate():
  n = 30
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  p = _plan(n, ops, g8)
  sw = p['sweeps']
  S = 16 * 2 ** n
  assert len(sw) == 3                           # 12 + 9 + 9 target bits
  assert sum(s['gates'] for s in sw) + p['noop_gates'] == 465
  # sweep 1: contiguous lanes, five register bits, one wave bit (a workgroup = 2 tiles = one super-tile)
  assert sw[0]['regpos'] == [6, 7, 8, 9, 10] and sw[0]['lanehi'] == [3, 4, 5] and sw[0]['wavepos'] == [11]
  assert sw[0]['dense_ops'] == 12
  # sweeps 2, 3: split-lane tiles (the lowest new bit in the wave id, then lanes, the highest in registers)
  # gathered into the second buffer; sweep 2 leaves its own bits on 3..11 and -- looking ahead -- sweep 3's
  # targets right above them
  assert sw[1]['wavepos'] == [12] and sw[1]['lanehi'] == [13, 14, 15] and sw[1]['regpos'] == [16, 17, 18, 19, 20]
  assert sw[2]['wavepos'] == [12] and sw[2]['lanehi'] == [13, 14, 15] and sw[2]['regpos'] == [16, 17, 18, 19, 20]
  assert all(s['swept_bytes'] == 2 * S for s in sw)            # one read + one write each
  # every H runs as an add-only butterfly; their scalars ride on a phase factor of the sweep (planner.h: fold_sink_)
  assert [s['butterfly_ops'] for s in sw] == [s['dense_ops'] for s in sw]
  # sweep 1 (contiguous tile) runs in place and stores the exchanged layout (one exchange); sweeps 2, 3
  # (split lanes) gather their tile and store it contiguously into the second buffer: the wave bit is
  # swapped into a register and the displaced bit once more, and nothing is swapped back
  assert [s['relayout'] for s in sw] == [0, 1, 1]
  # lane butterflies run on the VALU (planner.h choose_lane_paths): DPP partner fetch for lane bits 0..3,
  # v_permlane swaps for 4 and 5.  Sweep 1 (six lane targets, in place): three exchanges in, three back, one for
  # the wave bit; sweeps 2 and 3 (three lane targets, relayout: nothing is exchanged back)
  assert [s['lswap_ops'] for s in sw] == [7, 4, 4]
  assert [s['dpp_ops'] for s in sw] == [4, 1, 1]
  # minimal-touch bytes of BASELINE.md: 30 H x 2S + 435 CU1 x S/2 = 277.5 S
  assert sum(s['alg_bytes'] for s in sw) == int(277.5 * S)
  # lazy diagonal placement + tables: (almost) no per-gate loop terms left
  assert sw[0]['oterms'] <= 4 and sw[0]['groups'] <= 40


def test_lone_controlled_phase_moves_only_touched_amplitudes():
  n = 24
  sb = workloads.StreamBuilder()
  sb.applyc(gates.u1(0.3), 2, 5)            # bits 21 and 18: both above the lane bits
  p = _plan(n, *sb.arrays())
  (s,) = p['sweeps']
  assert s['fixed_ones'] == (1 << 21) | (1 << 18)
  assert s['swept_bytes'] == s['alg_bytes'] == 16 * 2 ** n // 2   # S/2, not 2S
  sb = workloads.StreamBuilder()
  sb.apply1(gates.tgate(), 3)
  (s,) = _plan(n, *sb.arrays())['sweeps']
  assert s['swept_bytes'] == 16 * 2 ** n                          # S


def test_commutation_rules_keep_order_where_it_matters():
  """H(a) X(b) H(a): the second H(a) may not overtake ... and a gate skipped for
  lack of register bits blocks later gates on its bits."""
  n = 20
  sb = workloads.StreamBuilder()
  for q in range(11):                      # 11 distinct high targets: only 9 fit one tile (3 lane + 5 register + 1 wave bit)
    sb.apply1(gates.hadamard(), q)
  sb.applyc(gates.pauli_x(), 5, 0)         # dense on qubit 0, control on qubit 5
  sb.apply1(gates.hadamard(), 5)
  p = _plan(n, *sb.arrays())
  assert len(p['sweeps']) == 2
  assert sum(s['gates'] for s in p['sweeps']) == 13
  # the simulation-driven choice keeps qubits 0 and 5 together: H(0) H(5) CX(5->0) H(5)
  # all run in the first sweep plus seven more H; the two left-over H gates follow
  assert p['sweeps'][0]['gates'] == 11 and p['sweeps'][1]['gates'] == 2
  assert len(p['sweeps'][0]['wavepos']) == 1
  sb = workloads.StreamBuilder()           # order must survive: S then H then S on one qubit
  for g in (gates.rx(0.3), gates.hadamard(), gates.rx(0.3)):
    sb.apply1(g, 3)
  p = _plan(n, *sb.arrays())
  assert len(p['sweeps']) == 1 and p['sweeps'][0]['dense_ops'] == 3 and p['sweeps'][0]['gates'] == 3


def test_x_gates_are_pushed_through_the_circuit():
  """Uncontrolled X gates become pending bit flips (planner.h propagate_x): X.H.X is ONE
  conjugated gate, the X.CU.X of "controlled by |0>" (circuit.py:166-169,207-215) is ONE gate
  with a zero-control, and a flip still pending at the end of the flush is executed there."""
  n = 20
  sb = workloads.StreamBuilder()
  for g in (gates.pauli_x(), gates.hadamard(), gates.pauli_x()):
    sb.apply1(g, 3)
  (s,) = _plan(n, *sb.arrays())['sweeps']
  assert s['dense_ops'] == 1 and s['gates'] == 3              # all three reference gates accounted for
  sb = workloads.StreamBuilder()
  sb.apply1(gates.pauli_x(), 2)
  sb.applyc(gates.ry(0.4), 2, 9)                              # controlled by qubit 2 being |0>
  sb.apply1(gates.pauli_x(), 2)
  (s,) = _plan(n, *sb.arrays())['sweeps']
  assert s['dense_ops'] == 1 and s['gates'] == 3
  sb = workloads.StreamBuilder()
  sb.apply1(gates.pauli_x(), 4)
  sb.apply1(gates.hadamard(), 7)
  (s,) = _plan(n, *sb.arrays())['sweeps']
  assert s['dense_ops'] == 2 and s['gates'] == 2              # the X itself runs at the end


def test_shard_bit_predicates_resolved_at_plan_time():
  n, nloc = 12, 10
  sb = workloads.StreamBuilder()
  sb.applyc(gates.hadamard(), 0, 5)        # control = shard bit 1
  sb.applyc(gates.u1(0.5), 5, 1)           # diagonal target = shard bit 0
  sb.apply1(gates.rz(0.2), 0)              # diagonal on shard bit 1
  ops, g8 = sb.arrays()
  for shard in range(4):
    p = _plan(n, ops, g8, shard=(nloc, shard))
    gates_run = sum(s['gates'] for s in p['sweeps'])
    want = (1 if shard & 2 else 0) + (1 if shard & 1 else 0) + 1
    assert gates_run == want and gates_run + p['noop_gates'] == 3


def test_supremacy_and_grover_streams_plan_completely():
  ops, g8 = workloads.supremacy_stream(30, 20, seed=0).arrays()
  p = _plan(30, ops, g8)
  assert sum(s['gates'] for s in p['sweeps']) + p['noop_gates'] == len(ops) == 342
  assert len(p['sweeps']) <= 8
  ops, g8 = workloads.grover_stream(10, [1, 0] * 5, iterations=1).arrays()
  p = _plan(20, ops, g8)
  assert sum(s['gates'] for s in p['sweeps']) + p['noop_gates'] == len(ops)


@pytest.mark.parametrize('name', ['qft30', 'sup30', 'grover34', 'sup16c64'])
def test_dry_flush_builds_the_device_op_buffers(name):
  """A flush of a planner-only handle goes through everything but the launches: planning, the device copies of ops and
  groups (handler numbers, factors folded into op headers, lane-table flags).  Guards the host-side passes over those
  buffers (a pass that read a header word after another had overwritten it crashed here first)."""
  lib = native.load()
  bw = 64 if name.endswith('c64') else 128
  if name.startswith('qft'):
    n, sb = 30, workloads.qft_stream(range(30))
  elif name.startswith('sup'):
    n = int(name[3:5])
    sb = workloads.supremacy_stream(n, 20, seed=0)
  else:
    n, sb = 34, workloads.grover_stream(17, [1, 0] * 8 + [1], iterations=1)
  ops, g8 = sb.arrays()
  g8 = np.ascontiguousarray(g8, dtype=np.float64)
  h = ctypes.c_void_p()
  native.check(lib.qh_create_dry(n, bw, ctypes.byref(h)))
  native.check(lib.qh_set_fusion(h, native.QH_FUSE_SWEEP))
  for rep in range(2):                 # (the second flush takes the cached plan)
    for k in range(len(ops)):
      gp = ctypes.cast(g8.ctypes.data + 64 * k, _dp)
      c, t = int(ops[k, 0]), int(ops[k, 1])
      native.check(lib.qh_apply1(h, t, gp) if c == workloads.NO_CTL else lib.qh_applyc(h, c, t, gp))
    native.check(lib.qh_flush(h))
  lib.qh_destroy(h)
