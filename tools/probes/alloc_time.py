#!/usr/bin/env python3
"""How long the state buffers of a 30-qubit handle take to allocate in a FRESH process (QH_ALLOC_DEBUG=1 prints each
buffer: size, kind, address, milliseconds), contiguous (default for 4-32 GiB) against plain hipMalloc, and what the first
flush then waits for.  usage: alloc_time.py [qubits]"""
import os
import subprocess
import sys
import time

n = sys.argv[1] if len(sys.argv) > 1 else '30'
child = r'''
import sys, time
sys.path.insert(0, %r)
from qcc_amd import device, native, workloads
n = int(sys.argv[1])
t0 = time.perf_counter()
st = device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP)
t1 = time.perf_counter()
st.init_basis(5); st.sync()
t2 = time.perf_counter()
ops, g8 = workloads.qft_stream(range(n)).arrays()
st.run_stream(ops, g8); st.sync()
t3 = time.perf_counter()
st.run_stream(ops, g8); st.sync()
t4 = time.perf_counter()
print('create %%.1f ms, init+sync %%.1f ms, first QFT %%.1f ms, second QFT %%.1f ms' %% ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
st.close()
''' % os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rep in range(3):
  for env in ({}, {'QH_ALLOC_CONTIG': '0'}, {'QH_PREALLOC': '0'}, {'QH_ALLOC_CONTIG': '0', 'QH_PREALLOC': '0'}):
    e = dict(os.environ, QH_ALLOC_DEBUG='1', **env)
    t = time.perf_counter()
    r = subprocess.run([sys.executable, '-c', child, n], env=e, capture_output=True, text=True)
    print(env, 'process %.2f s' % (time.perf_counter() - t))
    print('   ', (r.stdout.strip() + ' | ' + ' '.join(l for l in r.stderr.splitlines() if 'qh alloc' in l)))
