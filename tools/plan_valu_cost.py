#!/usr/bin/env python3
"""VALU wave-instructions per tile that the sweep islands execute for a planned circuit (no GPU needed): the plan of a named
workload through the dry handle (QH_PLAN_VERBOSE=1), every op and DIAG group priced as tools/gen_sweep_asm.py emits it
(complex128).  The sweeps of an op-heavy circuit are bound by VALU issue, so this is the figure a planner change
is judged by before it goes to the GPU; SQ_INSTS_VALU of the real launches is the check (profiles/r04/sq_counters_*.txt).
  usage: plan_valu_cost.py sup30 | sup30sK | qftNN | grover34   [-v]"""
import collections
import ctypes
import json
import os
import sys

os.environ['QH_PLAN_VERBOSE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from qcc_amd import native, workloads  # noqa: E402

OPF_REAL, OPF_BFLY, OPF_LANE_DPP, OPF_DEFER_C, OPF_USE_C = 4, 8, 128, 1, 2
DG_LTAB, DG_BITFAC = 1, 8


def plan(n, ops, g8):
  lib = native.load()
  h = ctypes.c_void_p()
  native.check(lib.qh_create_dry(n, 128, ctypes.byref(h)))
  native.check(lib.qh_set_fusion(h, native.QH_FUSE_SWEEP))
  g8 = np.ascontiguousarray(g8, dtype=np.float64)
  dp = ctypes.POINTER(ctypes.c_double)
  for k in range(len(ops)):
    gp = ctypes.cast(g8.ctypes.data + 64 * k, dp)
    c, t = int(ops[k, 0]), int(ops[k, 1])
    native.check(lib.qh_apply1(h, t, gp) if c == workloads.NO_CTL else lib.qh_applyc(h, c, t, gp))
  need = ctypes.c_uint64()
  lib.qh_plan_json(h, None, 0, ctypes.byref(need))
  buf = ctypes.create_string_buffer(need.value)
  lib.qh_plan_json(h, buf, need.value, None)
  lib.qh_destroy(h)
  return json.loads(buf.value.decode())


def workload(name):
  if name.startswith('sup'):
    n = int(name[3:5])
    seed = int(name[6:]) if name[5:6] == 's' else 0
    return (n,) + workloads.supremacy_stream(n, 20, seed=seed).arrays()
  if name == 'grover34':
    return (34,) + workloads.grover_stream(17, [1, 0] * 8 + [1], iterations=1).arrays()
  n = int(name[3:])
  return (n,) + workloads.qft_stream(range(n)).arrays()


def popc(x):
  return bin(x).count('1')


def price(sw, rb=5):
  nr = 1 << rb
  cost = collections.Counter()
  for op in sw['ops']:
    k, f = op['kind'], op['flags']
    if k == 0:
      if f & OPF_BFLY and f & (2048 | 4096):
        cost['register butterflies behind a pi/4 phase (fused)'] += 3 * nr
      elif f & OPF_BFLY:
        cost['register butterflies'] += 2 * nr
      elif f & OPF_REAL:
        cost['dense real (register)'] += 5 * nr
      else:
        cost['dense complex (register)'] += 10 * nr + 10
    elif k == 1:
      if f & OPF_LANE_DPP and f & OPF_BFLY:
        cost['lane butterflies by DPP'] += nr * (10 if op['tb'] == 2 else 6) + 14
      elif f & OPF_LANE_DPP:
        cost['lane real by DPP'] += nr * (12 if op['tb'] == 2 else 8) + 20
      elif f & OPF_BFLY:
        cost['lane butterflies (LDS shuffles)'] += 2 * nr + 8
      else:
        cost['lane dense (LDS shuffles)'] += (10 if not f & OPF_REAL else 4) * nr + 30
    elif k == 3:
      cost['lane <-> register exchanges'] += 2 * nr + 8
    elif k == 4:
      cost['wave <-> register exchanges (LDS)'] += 5
    elif k == 2:
      c_touched = c_sign = False
      for lane_mask, reg_mask, gf, ntab, noterms, re, im in op['groups']:
        slots = nr >> popc(reg_mask)
        general = (gf & DG_LTAB) or ntab or (noterms and not (re == 1.0 and im == 0.0 and False))
        sign = (re == -1.0 and im == 0.0) and not (gf & DG_LTAB) and not ntab
        if gf & DG_BITFAC:
          cost['factor trees'] += 124 + (11 if lane_mask or general else 0)
          continue
        if sign and not noterms or (sign and noterms):
          if reg_mask == 0:
            cost['sign on c'] += 6
            c_touched = True
            c_sign = c_sign or True
          else:
            cost['sign groups'] += 4 + 2 * slots
          continue
        pro = 0
        if general:
          pro = 4 + 4 * ntab + 4 * noterms + (4 if gf & DG_LTAB else 7)
        elif lane_mask:
          pro = 11
        if reg_mask == 0:
          cost['factors joining c'] += pro + 4
          c_touched = True
          c_sign = False
        else:
          cost['phase groups (%d-bit masks)' % popc(reg_mask)] += 4 * slots
          cost['group prologues'] += pro
      if c_touched and not (f & OPF_DEFER_C):
        cost['per-lane factor c on all slots'] += (2 * nr + 1) if c_sign else 4 * nr
  return cost


# Socket energy per wave-instruction by cost class, in nJ (profiles/r04/valu_power_per_instruction.txt: v_fma_f64 2.0, v_mul_f64 1.9,
# v_add_f64 1.5, v_mov_b64 1.3, v_xor_b32 1.0, v_mov_b32_dpp 0.75, v_mov_b32 0.7, v_permlane*_swap 1.2): the op streams of an
# op-heavy sweep run against the socket's power limit, so ENERGY, not the instruction count, is what a plan costs.
ENERGY = {
    'register butterflies': 1.5, 'register butterflies behind a pi/4 phase (fused)': 1.58, 'dense real (register)': 1.9,
    'dense complex (register)': 1.95, 'lane butterflies by DPP': 1.0, 'lane real by DPP': 1.1, 'lane butterflies (LDS shuffles)': 1.5,
    'lane dense (LDS shuffles)': 1.9, 'lane <-> register exchanges': 1.2, 'wave <-> register exchanges (LDS)': 1.0,
    'factor trees': 1.95, 'sign on c': 1.0, 'sign groups': 1.0, 'factors joining c': 1.9, 'group prologues': 1.0,
    'per-lane factor c on all slots': 1.95,
}


def energy(cost):
  return sum(v * ENERGY.get(k, 1.95 if k.startswith('phase groups') else 1.5) for k, v in cost.items())


def main():
  name = sys.argv[1] if len(sys.argv) > 1 else 'sup30'
  n, ops, g8 = workload(name)
  p = plan(n, ops, g8)
  total = collections.Counter()
  for i, sw in enumerate(p['sweeps']):
    c = price(sw, len(sw['regpos']))
    total.update(c)
    print(f'sweep {i}: {sw["gates"]} gates, {len(sw["ops"])} ops, {sw["groups"]} groups: {sum(c.values())} VALU instructions per tile')
    if '-v' in sys.argv:
      for k, v in c.most_common():
        print(f'    {v:6d}  {k}')
  print(f'{name}: {len(p["sweeps"])} sweeps, {sum(total.values())} VALU instructions per tile-circuit'
        f' = {sum(total.values()) * (1 << (n - 11)) / 1e9:.2f} G wave-instructions;'
        f' {energy(total) / 1e3:.1f} uJ per tile-circuit = {energy(total) * (1 << (n - 11)) / 1e9:.1f} J per circuit (price list)')
  for k, v in total.most_common():
    print(f'  {v:6d}  {k}')


if __name__ == '__main__':
  main()
