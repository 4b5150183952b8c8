#!/usr/bin/env python3
"""Times variants of the first QFT sweep (bits 0..10) to attribute sweep time
to op classes (run on the GPU box)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, gates, native, workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
st = device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP)
st.init_basis(5)
ops, g8 = workloads.qft_stream(range(n)).arrays()
st.run_stream(ops, g8)
st.sync()
isH = ops[:, 0] == workloads.NO_CTL
cb = np.where(isH, -1, n - 1 - ops[:, 0])
tb = n - 1 - ops[:, 1]


def run(name, sel, reps=3):
  st.sync()
  st.reset_stats()
  st.timer_begin()
  for _ in range(reps):
    st.run_stream(ops[sel], g8[sel])
    st.flush()
  ms = st.timer_end() / reps
  s = st.stats()
  print(json.dumps({'case': name, 'gates': int(sel.sum()), 'sweeps': s['sweeps'] // reps, 'ms': round(ms, 3)}))


first = ((tb <= 10) & isH) | (~isH & (cb <= 10))          # everything of sweep 1
run('sweep1 all', first)
run('H lane bits 0-5 only', isH & (tb <= 5))
run('H reg bits 6-10 only', isH & (tb >= 6) & (tb <= 10))
run('H bits 0-10', isH & (tb <= 10))
run('H 0-10 + cu1 with both bits in tile', (isH & (tb <= 10)) | (~isH & (cb <= 10) & (tb <= 10)))
run('H 0-10 + cu1 with outside target', (isH & (tb <= 10)) | (~isH & (cb <= 10) & (tb > 10)))
run('cu1 ctl<=10 outside targets only (diag-only sweep)', ~isH & (cb <= 10) & (tb > 10))
run('cu1 ctl lane, tgt reg', ~isH & (cb <= 5) & (tb >= 6) & (tb <= 10))
run('cu1 ctl lane, tgt lane', ~isH & (cb <= 5) & (tb <= 5))
H10 = isH & (tb <= 10)
run('H + cu1 lane-lane (15)', H10 | (~isH & (cb <= 5) & (tb <= 5)))
run('H + cu1 lane->reg (30)', H10 | (~isH & (cb <= 5) & (tb >= 6) & (tb <= 10)))
run('H + cu1 reg-reg (10)', H10 | (~isH & (cb >= 6) & (cb <= 10) & (tb <= 10)))
run('H + cu1 lane->outside (114)', H10 | (~isH & (cb <= 5) & (tb > 10)))
run('H + cu1 reg->outside (95)', H10 | (~isH & (cb >= 6) & (cb <= 10) & (tb > 10)))
