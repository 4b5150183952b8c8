#!/usr/bin/env python3
"""Is there a K-sweep tiling?  The exact answer for a circuit, as an integer program (scipy.optimize.milp / HiGHS; CPU, seconds).

A plan of K sweeps is a labelling of the circuit's DENSE gates with levels 0..K-1 that never decreases along a dependency
(diagonal gates ride along: they need no tile bit), and the tile of sweep s is the set of index bits with a gate on level s: at
most `cap` of them besides the line bits (index bits 0-2), which every tile holds anyway.  The dependency graph is the planner's
own (qh_plan_json with QH_PLAN_DAG=1: planner.h build_dag -- what search_levels walks on).  Variables: x[g,s] = "gate g is on a
level <= s" (monotone in s and along every edge), y[b,s] = "bit b is in tile s"; sum_b y[b,s] <= cap.  search_levels looks for
the same object by local search inside a time budget; this tool says what the minimum IS (profiles/r06/level_search.txt).

  usage: tiling_milp.py WORKLOAD K CAP [TIME_LIMIT_S]     WORKLOAD as tools/plan_valu_cost.py: sup30 | sup30sK | qftNN | grover34
         CAP = 8 / 9 / 10 movable bits for tiles with no / one / two wave bits (complex128)"""
import os
import sys
import time

os.environ['QH_PLAN_DAG'] = '1'
os.environ['QH_PLAN_SEARCH'] = '0'
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
from scipy.optimize import Bounds, LinearConstraint, milp  # noqa: E402
from scipy.sparse import lil_matrix  # noqa: E402

import plan_valu_cost  # noqa: E402


def solve(dag, K, cap, time_limit=600):
  tb, edges, line = dag['target_bit'], dag['edges'], dag['line_bits']
  ng = len(tb)
  bits = sorted({b for b in tb if b >= line})
  xi = {(g, s): g * (K - 1) + s for g in range(ng) for s in range(K - 1)}
  nv = ng * (K - 1)
  yi = {}
  for b in bits:
    for s in range(K):
      yi[b, s] = nv
      nv += 1
  rows, hi = [], []

  def X(g, s):
    return (None, 0) if s < 0 else (None, 1) if s >= K - 1 else (xi[g, s], None)

  def le(a, b):                                  # a <= b, each (index, constant)
    t, rhs = [], 0
    if a[0] is not None: t.append((a[0], 1))
    else: rhs -= a[1]
    if b[0] is not None: t.append((b[0], -1))
    else: rhs += b[1]
    if t:
      rows.append(t)
      hi.append(rhs)

  for g in range(ng):
    for s in range(K - 2):
      le(X(g, s), X(g, s + 1))
  for u, v in edges:                             # level(u) <= level(v)
    for s in range(K - 1):
      le(X(v, s), X(u, s))
  for g in range(ng):
    if tb[g] < line:
      continue
    for s in range(K):                            # y[b,s] >= x[g,s] - x[g,s-1]
      t, rhs = [(yi[tb[g], s], -1)], 0
      i1, c1 = X(g, s)
      i0, c0 = X(g, s - 1)
      if i1 is not None: t.append((i1, 1))
      else: rhs -= c1
      if i0 is not None: t.append((i0, -1))
      else: rhs += c0
      rows.append(t)
      hi.append(rhs)
  for s in range(K):
    rows.append([(yi[b, s], 1) for b in bits])
    hi.append(cap)
  M = lil_matrix((len(rows), nv))
  for r, t in enumerate(rows):
    for i, c in t:
      M[r, i] += c
  t0 = time.time()
  res = milp(np.zeros(nv), constraints=LinearConstraint(M.tocsr(), -np.inf, hi), integrality=np.ones(nv), bounds=Bounds(0, 1),
             options={'time_limit': time_limit, 'disp': False})
  tiles = None
  if res.status == 0 and res.x is not None:
    x = np.round(res.x)
    level = [K - 1 - int(sum(x[xi[g, s]] for s in range(K - 1))) for g in range(ng)]
    tiles = [sorted({tb[g] for g in range(ng) if level[g] == s and tb[g] >= line}) for s in range(K)]
  return res.status, time.time() - t0, tiles


def main():
  name, K, cap = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
  limit = float(sys.argv[4]) if len(sys.argv) > 4 else 600
  n, ops, g8 = plan_valu_cost.workload(name)
  dag = plan_valu_cost.plan(n, ops, g8)['dag']
  status, dt, tiles = solve(dag, K, cap, limit)
  movable = len({b for b in dag['target_bit'] if b >= dag['line_bits']})
  print(f'{name}: {len(dag["target_bit"])} dense gates on {movable} movable bits, {len(dag["edges"])} dependencies; K = {K}, cap = {cap}: '
        + ('FEASIBLE' if status == 0 else 'INFEASIBLE' if status == 2 else f'undecided (status {status})') + f' ({dt:.1f} s)')
  if tiles:
    for s, t in enumerate(tiles):
      print(f'  sweep {s}: {len(t)} index bits', t)
    print('  QH_PLAN_TILES=%d:%s' % (max(0, cap - dag['cap_per_wave_bits'][0]), ';'.join(','.join(map(str, t)) for t in tiles)))


if __name__ == '__main__':
  main()
