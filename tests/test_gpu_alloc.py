"""State buffers of 4-32 GiB are requested as physically contiguous VRAM (engine.hip alloc_state_buffer; DESIGN 7
"placement"); QH_ALLOC_CONTIG=0 / 1 force either kind.  The result of a circuit must not depend on it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from qcc_amd import device, native, workloads
n = 28                                            # 4 GiB per buffer: the smallest size the default asks contiguous
ops, g8 = workloads.qft_stream(range(n)).arrays()
with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
  st.init_basis(0x2CB9A5E3 & ((1 << n) - 1))
  st.run_stream(ops, g8); st.flush()
  idx = np.array([0, 1, 12345, (1 << n) - 1], dtype=np.uint64)
  print('RESULT', repr(st.norm2()), ' '.join(repr(complex(st.amplitude(int(k)))) for k in idx))
''' % ROOT


def _run(**env):
  e = dict(os.environ, QH_ALLOC_DEBUG='1')
  e.update(env)
  r = subprocess.run([sys.executable, '-c', SCRIPT], capture_output=True, text=True, timeout=600, env=e)
  assert r.returncode == 0, r.stderr[-2000:]
  kinds = [ln.split()[4] for ln in r.stderr.splitlines() if ln.startswith('[qh alloc 4096 MiB')]
  res = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')]
  assert len(res) == 1
  return kinds, res[0]


def test_default_asks_for_contiguous_buffers_and_the_switch_turns_it_off():
  kinds, res = _run()
  assert kinds and set(kinds) <= {'contiguous', 'plain'}      # (plain only if the driver had no contiguous range)
  kinds0, res0 = _run(QH_ALLOC_CONTIG='0')
  assert kinds0 and set(kinds0) == {'plain'}
  assert res.split()[2:] == res0.split()[2:]                  # bit-identical amplitudes either way
  assert abs(float(res.split()[1]) - 1) < 1e-12 and abs(float(res0.split()[1]) - 1) < 1e-12
