#!/usr/bin/env python3
"""Pre-flight check of a multi-GPU run WITHOUT any GPU: plans the n-qubit QFT (or supremacy) for every rank of a P-rank
run through the routing and the planner a real run uses (qcc_amd.sharded.DryShard: planner-only engine handles) and
prints, per exchange, how it would be cut -- slabs, rounds, chunk size, packed / direct, staging bytes, sweeps in
front of it -- and whether all ranks agree (they must: RCCL send/recv counts and landing offsets follow from it;
engine.hip compares the signature before data moves).

  python tools/plan_sharded.py 36 8            # BASELINE config 5
  python tools/plan_sharded.py 34 2 --reps 3 --workload qft
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcc_amd import sharded, workloads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('qubits', type=int)
ap.add_argument('ranks', type=int)
ap.add_argument('--reps', type=int, default=2, help='times the circuit is submitted (a loop sees other layouts from the 2nd step on)')
ap.add_argument('--workload', default='qft', choices=['qft', 'supremacy'])
args = ap.parse_args()
n, world = args.qubits, args.ranks
ops, g8 = (workloads.qft_stream(range(n)) if args.workload == 'qft' else workloads.supremacy_stream(n, 20, seed=0)).arrays()
records = []
for rank in range(world):
  sh = sharded.DryShard(n, world, rank)
  marks = []
  for _ in range(args.reps):
    sh.run_stream(ops, g8)
    sh.flush()
    marks.append((sh.stats()['sweeps'], sh.exchanges))
  records.append({'geometries': sh.geometries, 'marks': marks, 'bitmap': list(sh.perm)})
  sh.close()
agree = all(r == records[0] for r in records)
nloc = n - (world.bit_length() - 1)
print(f'{args.workload} on {n} qubits, {world} ranks, 2^{nloc} amplitudes ({(16 << nloc) >> 30} GiB) per rank, {len(ops)} gates per step')
prev = (0, 0)
for step, (sweeps, xch) in enumerate(records[0]['marks']):
  print(f'  step {step}: {sweeps - prev[0]} sweeps, {xch - prev[1]} exchange(s)')
  prev = (sweeps, xch)
for k, g in enumerate(records[0]['geometries']):
  print(f'  exchange {k}: {g["sweeps_before"]} sweeps in front (last one {"cut into slabs" if g["last_sweep_split"] else "whole"}), '
        f'{g["slabs"]} slabs x {g["rounds_per_slab"]} rounds x {g["peers"]} peers x 2^{g["chunk_bits"]} amplitudes, '
        f'{"packed (gather / scatter kernels)" if g["packed"] else "direct (sent from where they lie)"}, '
        f'staging {g["staging_bytes"] / 2**30:.2f} GiB, block bits {g["block_bits"]:#x}, slab bits {g["slab_mask"]:#x}, '
        f'signature {g["signature"]:016x}')
print('all ranks agree on every exchange:', agree)
if not agree:
  for r, rec in enumerate(records):
    if rec != records[0]:
      print(f'  rank {r} differs:', json.dumps(rec['geometries']))
  sys.exit(1)
