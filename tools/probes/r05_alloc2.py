#!/usr/bin/env python3
"""Second buffer plain vs contiguous (QH_ALLOC_CONTIG=2 vs default vs 0): steady 30-qubit QFT step time and allocation times
over fresh processes.  usage: r05_alloc2.py [rounds]"""
import os
import subprocess
import sys

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
child = r'''
import sys, time
sys.path.insert(0, %r)
from qcc_amd import device, native, workloads
n = 30
if len(sys.argv) > 1 and sys.argv[1] == 'churn':      # what bench.py does before its single-shot circuits: big states come and go
  for q in (30, 33, 34, 30):
    with device.DeviceState(q, 128, fusion=native.QH_FUSE_SWEEP) as big:
      big.init_basis(1)
      ops_, g8_ = workloads.qft_stream(range(q)).arrays()
      if q <= 33:
        big.run_stream(ops_, g8_)
      big.sync()
st = device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP)
st.init_basis(5)
ops, g8 = workloads.qft_stream(range(n)).arrays()
for _ in range(3):
  st.run_stream(ops, g8); st.flush()
st.sync(); st.timer_lap()
for _ in range(12):
  st.run_stream(ops, g8); st.flush(); st.timer_lap()
laps = st.timer_laps(); st.sync()
laps = sorted(laps)
print('median %%.3f min %%.3f' %% (laps[len(laps)//2], laps[0]))
st.close()
''' % root
for r in range(rounds):
  for churn in ((), ('churn',)):
    if churn and r >= 3:
      continue
    for env in ({}, {'QH_ALLOC_CONTIG': '2'}, {'QH_ALLOC_CONTIG': '0'}):
      e = dict(os.environ, QH_ALLOC_DEBUG='1', **env)
      p = subprocess.run([sys.executable, '-c', child, *churn], env=e, capture_output=True, text=True)
      allocs = ' '.join(l.split(' at ')[0].replace('[qh alloc ', '') + ' ' + l.split(' in ')[1].rstrip(']') for l in p.stderr.splitlines() if 'qh alloc 16384' in l)
      print(r, 'churn' if churn else 'fresh', env or 'default', p.stdout.strip(), '|', allocs[-160:], flush=True)
