#!/usr/bin/env python3
"""Round 6, VERDICT r05 #1(a): socket power and issue rate of the FP64 matrix pipe (tools/membench/mfmapower) beside the VALU
kinds it would replace, at k_sweep's occupancy; the same sampling as round 4's VALU price list (rocm-smi, median of 5).  GPU box.
  python tools/probes/r06_mfmapower.py > gpurun_out/r06_mfma.txt"""
import json
import os
import subprocess
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(R, 'tools', 'membench', 'mfmapower')


def smi():
  r = subprocess.run(['rocm-smi', '-d', '0', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10)
  card = next(iter(json.loads(r.stdout).values()))
  p = float(card['Current Socket Graphics Package Power (W)'])
  s = card.get('sclk clock speed:', '(0Mhz)')
  return p, int(s.strip('()').replace('Mhz', ''))


if len(sys.argv) == 1:
  print(subprocess.run([EXE, 'layout'], capture_output=True, text=True).stdout)
idle = smi()
print(f'idle: {idle[0]:.0f} W, sclk {idle[1]} MHz')
MODES = sys.argv[1:] or ['fma', 'add', 'mfma4', 'mfma16', 'dppbf', 'grp', 'grpmix', 'salu', 'smem']
for mode in MODES:
  p = subprocess.Popen([EXE, mode, '8'], stdout=subprocess.PIPE, text=True)
  time.sleep(2.5)
  samples = []
  while p.poll() is None and len(samples) < 5:
    samples.append(smi())
    time.sleep(0.6)
  out = p.communicate()[0].strip()
  pw = sorted(s[0] for s in samples)[len(samples) // 2] if samples else 0
  ck = sorted(s[1] for s in samples)[len(samples) // 2] if samples else 0
  try:
    rate = float(out.split()[1])
  except (IndexError, ValueError):
    rate = 0.0
  print(f'{mode:7s} {pw:6.0f} W  sclk {ck} MHz  {out}  -> {(pw - idle[0]) / rate * 1e9 if rate else 0:.2f} nJ per unit above idle')
