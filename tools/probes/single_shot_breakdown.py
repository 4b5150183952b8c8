import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
os.environ['QH_PLAN_CACHE'] = '0'
import numpy as np
from qcc_amd.lib import circuit, tensor
tensor.set_tensor_width(128)
n = 30
for rep in range(4):
  qc = circuit.qc('single-shot')
  reg = qc.reg(n, rep + 1)
  qc.maxprob()
  t0 = time.perf_counter()
  qc.qft(reg)
  t1 = time.perf_counter()
  qc.sync() if hasattr(qc, 'sync') else None
  t2 = time.perf_counter()
  bits, p = qc.maxprob()
  t3 = time.perf_counter()
  print('rep %d: qft (python) %.2f ms, sync (drain + flush + wait) %.2f ms, maxprob %.2f ms' % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
  del qc
