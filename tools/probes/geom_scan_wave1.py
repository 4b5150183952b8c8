#!/usr/bin/env python3
"""9 target bits = 3 lane-high + 5 register + 1 wave bit: all role choices for a range (GPU box)."""
import itertools
import subprocess
import sys

exe = 'tools/membench/geomsweep'
for start in [int(a) for a in sys.argv[1:]] or [12, 21]:
  bits = list(range(start, start + 9))
  geoms = []
  for w in bits:
    rest = [b for b in bits if b != w]
    for lanes in itertools.combinations(rest, 3):
      if max(lanes) > 27:
        continue
      regs = [b for b in rest if b not in lanes]
      geoms.append(list(lanes) + regs + [w])
  res = []
  for i in range(0, len(geoms), 64):
    out = subprocess.run([exe, '30'] + [','.join(map(str, g)) + ':3' for g in geoms[i:i + 64]], capture_output=True, text=True).stdout
    res += [(float(l.split()[1]), l.split()[0]) for l in out.splitlines() if ' ms ' in l]
  res.sort()
  print(start, 'n', len(res), 'best', res[:6], 'worst', res[-2:])
  d = [r for r in res if r[1].startswith(f'{start+1},{start+2},{start+3},') and r[1].split(':')[0].endswith(f',{start}')]
  print('  current rule (wave lowest, lanes next):', d)
