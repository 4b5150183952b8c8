#!/usr/bin/env python3
"""Cost of wave-bit gates (OP_WSWAP): H on sets of index bits, one fused sweep each."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qcc_amd import device, gates, native  # noqa: E402

n = 30
st = device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP)
st.init_basis(5)


def run(name, bits, reps=3):
  def go():
    for b in bits:
      st.apply_bits(0, b, gates.hadamard())
    st.flush()
  go(); st.sync(); st.reset_stats(); st.timer_begin()
  for _ in range(reps):
    go()
  ms = st.timer_end() / reps
  s = st.stats()
  print(json.dumps({'case': name, 'bits': list(bits), 'sweeps': s['sweeps'] // reps, 'ms': round(ms, 3)}))


run('5 regs high', range(23, 28))
run('5 regs + 2 waves high', range(23, 30))
run('5 regs + 1 wave', range(23, 29))
run('contig lanes + regs low (3..10)', range(3, 11))
run('3..12 (lanes, regs, 2 waves)', range(3, 13))
run('13..22 split lanes + regs + waves', range(13, 23))
run('13..20 split lanes + regs', range(13, 21))
