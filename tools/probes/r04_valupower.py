#!/usr/bin/env python3
"""Socket power while ONE kind of VALU instruction runs on every SIMD (tools/membench/valupower): the energy price list
behind "the sweeps are bound by the socket's power limit" (DESIGN 4.5).  GPU box."""
import json
import os
import subprocess
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def smi():
  r = subprocess.run(['rocm-smi', '-d', '0', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10)
  card = next(iter(json.loads(r.stdout).values()))
  p = float(card['Current Socket Graphics Package Power (W)'])
  s = card.get('sclk clock speed:', '(0Mhz)')
  return p, int(s.strip('()').replace('Mhz', ''))


idle = smi()
print(f'idle: {idle[0]:.0f} W, sclk {idle[1]} MHz')
for mode in ('fma', 'add', 'mul', 'mov64', 'mov32', 'dpp', 'xor', 'swap32'):
  p = subprocess.Popen([os.path.join(R, 'tools', 'membench', 'valupower'), mode, '8'], stdout=subprocess.PIPE, text=True)
  time.sleep(2.5)
  samples = []
  while p.poll() is None and len(samples) < 5:
    samples.append(smi())
    time.sleep(0.6)
  out = p.communicate()[0].strip()
  pw = sorted(s[0] for s in samples)[len(samples) // 2] if samples else 0
  ck = sorted(s[1] for s in samples)[len(samples) // 2] if samples else 0
  rate = float(out.split()[1]) if out.startswith('wave') else 0.0
  print(f'{mode:7s} {pw:6.0f} W  sclk {ck} MHz  {out}  -> {(pw - idle[0]) / rate * 1e9 if rate else 0:.2f} nJ per wave-instruction above idle')
