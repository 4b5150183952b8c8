#!/usr/bin/env python3
"""CPU-side fuzzing of the sweep planner: plan (qh_plan_export) -> tests/plan_interp.py -> oracle.

usage: fuzz_planner_cpu.py SEED SECONDS      (test tool; runs without a GPU)

Random circuits (tests/test_planner_semantics_cpu.py builds them: every gate class, multi-controls,
non-unitary operators, shard bits) are planned in every plan shape the engine can be switched to;
the exported plan is executed with NumPy and compared with the oracle."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib, plan_interp  # noqa: E402
from tests.test_planner_semantics_cpu import (ENVS, GEOMETRY_KEYS, _oracle_apply, _planned, _shard_variant_stream,  # noqa: E402
                                               _stream)

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
orc = oracle_lib.load()
t0 = time.time()
cases = fails = 0
while time.time() - t0 < budget:
  env = ENVS[cases % len(ENVS)]
  for k in [k for k in os.environ if k.startswith('QH_')]:
    del os.environ[k]
  os.environ.update(env)
  rng = np.random.default_rng(seed * 7919 + cases)
  n = int(rng.integers(10, 17))
  gshard = int(rng.integers(0, 4)) if rng.random() < 0.4 else 0
  if cases % 2 and os.environ.get('QH_PLAN_SEARCH_STEPS') is None:
    os.environ['QH_PLAN_SEARCH_STEPS'] = '300000'        # the level search too (small states get no budget by default)
  n = max(n, 10 + gshard)
  # sharded cases: half of them with plenty of rank-dependent gates (ghosts on some ranks, planner.h)
  stream = (_shard_variant_stream(rng, n, int(rng.integers(10, 300)), gshard) if gshard and rng.random() < 0.5
            else _stream(rng, n, int(rng.integers(10, 400)), gshard))
  psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
  psi = (psi / np.linalg.norm(psi)).astype(np.complex128)
  want = psi.copy()
  _oracle_apply(orc, want, n, stream)
  nloc = n - gshard
  got = np.empty_like(psi)
  try:
    geom0 = None
    for shard in range(1 << gshard):
      part = psi[shard << nloc: (shard + 1) << nloc].copy()
      sweeps = _planned(n, nloc, shard, stream, bw=int(os.environ.get('FUZZ_BW', 128 if cases % 3 else 64)))
      geom = [[(k, sp[k]) for k in GEOMETRY_KEYS] for sp in sweeps]
      geom0 = geom if geom0 is None else geom0
      assert geom == geom0, f'shard {shard} plans another geometry than shard 0'
      plan_interp.run_plan(part, sweeps, nloc, shard)
      got[shard << nloc: (shard + 1) << nloc] = part
    err = float(np.max(np.abs(got - want)))
  except AssertionError as e:
    err = float('inf')
    print('ASSERT', e)
  if not err < 1e-10:
    fails += 1
    print('FAIL', cases, env, n, gshard, len(stream), err)
  cases += 1
print('cases', cases, 'fails', fails)
sys.exit(1 if fails else 0)
