#!/usr/bin/env python3
"""Scan tile geometries with tools/membench/geomsweep (GPU box): for every contiguous
8-bit range [s, s+8) and every choice of 3 lane bits among them, the RMW sweep rate."""
import itertools
import json
import subprocess
import sys

exe = 'tools/membench/geomsweep'
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 30
geoms = []
for s in range(8, nb - 7):
  bits = list(range(s, s + 8))
  for lanes in itertools.combinations(bits, 3):
    if max(lanes) > 27:
      continue
    regs = [b for b in bits if b not in lanes]
    geoms.append(list(lanes) + regs)
# last-sweep shapes: top t bits + fillers from low bits, contiguous lanes 3,4,5
for top in (3, 4, 5):
  tb = list(range(nb - top, nb))
  for fill in ([6, 7], [8, 9], [10, 11], [12, 13], [14, 15], [16, 17], [18, 19], [20, 21]):
    regs = (fill + tb)[-5:] if top < 5 else tb
    geoms.append([3, 4, 5] + sorted(regs))
res = []
for i in range(0, len(geoms), 64):
  chunk = geoms[i:i + 64]
  out = subprocess.run([exe, str(nb)] + [','.join(map(str, g)) for g in chunk], capture_output=True, text=True).stdout
  for line in out.splitlines():
    f = line.split()
    if len(f) >= 4 and f[2] == 'ms':
      res.append({'geom': f[0], 'ms': float(f[1])})
json.dump(res, open('gpurun_out/geom_scan.json', 'w'))
by = {}
for r in res:
  g = list(map(int, r['geom'].split(',')))
  key = min(g[3:] + g[:3]) if g[:3] != [3, 4, 5] else 'last:' + ','.join(map(str, g[3:]))
  by.setdefault(key, []).append((r['ms'], r['geom']))
for k, v in by.items():
  v.sort()
  print(k, 'best', v[0], 'worst', v[-1], 'default(low3 lanes)', [x for x in v if isinstance(k, int) and x[1].startswith(f'{k},{k+1},{k+2},')])
