#!/usr/bin/env python3
"""Scan role assignments (3 lane-high, 5 register, 2 wave bits) of a 10-bit target range with
tools/membench/geomsweep (GPU box).  usage: geom_scan_waves.py START [NBITS]"""
import itertools
import subprocess
import sys

exe = 'tools/membench/geomsweep'
start = int(sys.argv[1]) if len(sys.argv) > 1 else 13
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bits = list(range(start, start + 10))
geoms = []
for waves in itertools.combinations(bits, 2):
  rest = [b for b in bits if b not in waves]
  for lanes in itertools.combinations(rest, 3):
    if max(lanes) > 27:
      continue
    regs = [b for b in rest if b not in lanes]
    geoms.append(list(lanes) + regs + list(waves))
res = []
for i in range(0, len(geoms), 64):
  chunk = geoms[i:i + 64]
  out = subprocess.run([exe, str(nb)] + [','.join(map(str, g)) + ':3' for g in chunk], capture_output=True, text=True).stdout
  for line in out.splitlines():
    f = line.split()
    if len(f) >= 4 and f[2] == 'ms':
      res.append((float(f[1]), f[0]))
res.sort()
print('n', len(res))
for r in res[:15]:
  print(r)
print('...')
for r in res[-5:]:
  print(r)
d = [r for r in res if r[1].startswith(f'{start},{start+1},{start+2},') and r[1].split(':')[0].endswith(f'{start+8},{start+9}')]
print('default (lowest lanes, highest waves):', d)
