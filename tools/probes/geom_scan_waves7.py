#!/usr/bin/env python3
"""Contiguous lanes (3,4,5) + 5 register + 2 wave bits over a 7-bit target range: all 21 role choices."""
import itertools
import subprocess
import sys

exe = 'tools/membench/geomsweep'
for start in [int(a) for a in sys.argv[1:]] or [6, 13, 20, 23]:
  bits = list(range(start, start + 7))
  geoms = []
  for waves in itertools.combinations(bits, 2):
    regs = [b for b in bits if b not in waves]
    geoms.append([3, 4, 5] + regs + list(waves))
  out = subprocess.run([exe, '30'] + [','.join(map(str, g)) + ':3' for g in geoms], capture_output=True, text=True).stdout
  res = sorted((float(l.split()[1]), l.split()[0]) for l in out.splitlines() if ' ms ' in l)
  print(start, 'best', res[:3], 'worst', res[-2:], 'highest-waves', [r for r in res if r[1].split(':')[0].endswith(f'{start+5},{start+6}')])
