"""A short run of the differential fuzzer (tools/fuzz_parity.py): random circuits of random
shape through the fused path vs the CPU oracle, default plan and the alternative plan shapes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('env', [{}, {'QH_WAVE_BITS': '2', 'QH_LANE_VALU': '2'}, {'QH_WAVE_BITS': '0', 'QH_SWEEP_RB': '4', 'QH_PROPAGATE_X': '0'},
                                 {'QH_RELAYOUT': '0'}, {'QH_SEATS': '2', 'QH_WAVE_BITS': '1'},
                                 {'QH_ROT_FUSE': '0', 'QH_SEATS': '0'}])
def test_fuzz_fused_vs_oracle(env):
  e = dict(os.environ)
  e.update(env)
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_parity.py'), '8', '4242'], env=e, cwd=ROOT,
                     capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
  assert '"failures": 0' in r.stdout
