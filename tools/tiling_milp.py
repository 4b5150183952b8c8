#!/usr/bin/env python3
"""Is there a K-sweep tiling?  The exact answer for a circuit, as an integer program (scipy.optimize.milp / HiGHS; CPU, seconds).

A plan of K sweeps is a labelling of the circuit's DENSE gates with levels 0..K-1 that never decreases along a dependency
(diagonal gates ride along: they need no tile bit), and the tile of sweep s is the set of qubits with a gate on level s: at most
`cap` of them besides the three qubits of the 128-byte line (index bits 0-2), which every tile holds anyway.  Variables:
x[g,s] = "gate g is on a level <= s" (monotone in s and along each qubit's chain), y[q,s] = "qubit q is in tile s";
sum_q y[q,s] <= cap.  planner.h's search_levels looks for the same object by local search inside a time budget; this tool
says what the minimum IS (profiles/r06/level_search.txt: supremacy-30, depth 20, seeds 0-23).

  usage: tiling_milp.py SEED K CAP         (supremacy-30 depth 20; CAP = 9 for tiles with one wave bit, 10 with two)"""
import os
import sys
import time

import numpy as np
from scipy.optimize import Bounds, LinearConstraint, milp
from scipy.sparse import lil_matrix

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcc_amd import workloads  # noqa: E402


def model(n, depth, seed):
  """Dense gates per qubit and the cross constraints a CZ leaves: (a, i, b, j) = the i-th dense gate of qubit a (1-based) must
  not be on a later level than the j-th of qubit b."""
  ops, g8 = workloads.supremacy_stream(n, depth, seed=seed).arrays()
  g = np.asarray(g8, dtype=np.float64).reshape(-1, 8)
  cnt = [0] * n
  pend = []
  for k in range(len(ops)):
    c, t = int(ops[k, 0]), int(ops[k, 1])
    if not g[k, 2:6].any():                      # diagonal: T, CZ
      if c != workloads.NO_CTL:
        pend.append((c, cnt[c], t, cnt[t]))
    else:
      cnt[t] += 1
  cons = set()
  for a, ia, b, ib in pend:
    if ia >= 1 and ib + 1 <= cnt[b]:
      cons.add((a, ia, b, ib + 1))
    if ib >= 1 and ia + 1 <= cnt[a]:
      cons.add((b, ib, a, ia + 1))
  return cnt, sorted(cons)


def solve(n, cnt, cons, K, cap, free, time_limit=600):
  xi, yi, nv = {}, {}, 0
  for q in range(n):
    for j in range(cnt[q]):
      for s in range(K - 1):
        xi[q, j, s] = nv
        nv += 1
  for q in range(n):
    if q not in free:
      for s in range(K):
        yi[q, s] = nv
        nv += 1
  rows, hi = [], []

  def X(q, j, s):
    return (None, 0) if s < 0 else (None, 1) if s >= K - 1 else (xi[q, j, s], None)

  def le(a, b):                                  # a <= b, each (index, constant)
    t, rhs = [], 0
    if a[0] is not None: t.append((a[0], 1))
    else: rhs -= a[1]
    if b[0] is not None: t.append((b[0], -1))
    else: rhs += b[1]
    if t:
      rows.append(t)
      hi.append(rhs)

  for q in range(n):
    for j in range(cnt[q]):
      for s in range(K - 2):
        le(X(q, j, s), X(q, j, s + 1))
      if j + 1 < cnt[q]:
        for s in range(K - 1):
          le(X(q, j + 1, s), X(q, j, s))
  for a, ja, b, jb in cons:
    for s in range(K - 1):
      le(X(b, jb - 1, s), X(a, ja - 1, s))
  for q in range(n):
    if q in free:
      continue
    for j in range(cnt[q]):
      for s in range(K):                          # y[q,s] >= x[q,j,s] - x[q,j,s-1]
        t, rhs = [(yi[q, s], -1)], 0
        i1, c1 = X(q, j, s)
        i0, c0 = X(q, j, s - 1)
        if i1 is not None: t.append((i1, 1))
        else: rhs -= c1
        if i0 is not None: t.append((i0, -1))
        else: rhs += c0
        rows.append(t)
        hi.append(rhs)
  for s in range(K):
    rows.append([(yi[q, s], 1) for q in range(n) if q not in free])
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
    tiles = [[q for q in range(n) if q not in free and x[yi[q, s]] > 0.5] for s in range(K)]
  return res.status, time.time() - t0, tiles


def main():
  seed, K, cap = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
  n = 30
  cnt, cons = model(n, 20, seed)
  status, dt, tiles = solve(n, cnt, cons, K, cap, {n - 1, n - 2, n - 3})
  print(f'supremacy-30 depth 20 seed {seed}: {sum(cnt)} dense gates, {len(cons)} cross constraints; K = {K}, cap = {cap}: '
        + ('FEASIBLE' if status == 0 else 'INFEASIBLE' if status == 2 else f'status {status}') + f' ({dt:.1f} s)')
  if tiles:
    for s, t in enumerate(tiles):
      print(f'  sweep {s}: index bits', sorted(n - 1 - q for q in t))
    print('  QH_PLAN_TILES=%d:%s' % (cap - 8, ';'.join(','.join(str(n - 1 - q) for q in sorted(t, reverse=True)) for t in tiles)))


if __name__ == '__main__':
  main()
