import os, sys, gc
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qcc_amd.lib import circuit, tensor, backend
tensor.set_tensor_width(128)
n = 22
for rep in range(3):
  qc = circuit.qc('x')
  reg = qc.reg(n, rep + 1)
  qc.qft(reg)
  print('rep', rep, 'maxprob', qc.maxprob()[1], 'dev', hex(id(qc._dev)), 'pool', {k: len(v) for k, v in backend._pool.items()}, flush=True)
  print('  referrers of qc:', len(gc.get_referrers(qc)), [type(r).__name__ for r in gc.get_referrers(qc)][:6])
  del qc
  print('  after del: pool', {k: len(v) for k, v in backend._pool.items()}, 'garbage', gc.collect(), 'after collect', {k: len(v) for k, v in backend._pool.items()}, flush=True)
