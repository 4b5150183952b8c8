#!/usr/bin/env python3
"""Round 6: Joules per sweep -- the energy roofline of k_sweep.  Loops (a) a sweep with an EMPTY op stream (one T gate on a 30-qubit
state: the tile stream of a contiguous in-place sweep and nothing else), (b) the same through a gather / relayout sweep (one H on
index bit 20), (c) the 30-qubit QFT, (d) supremacy-30, and samples the socket power (rocm-smi, ~2 Hz) and the shader clock beside the
per-launch HIP-event time: J per sweep = socket W x ms per sweep.  GPU box.  python tools/probes/r06_energy.py > gpurun_out/r06_energy.txt"""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from qcc_amd import device, gates, native, workloads  # noqa: E402


def smi():
  r = subprocess.run(['rocm-smi', '-d', '0', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10)
  card = next(iter(json.loads(r.stdout).values()))
  p = float(card['Current Socket Graphics Package Power (W)'])
  s = card.get('sclk clock speed:', '(0Mhz)')
  return p, int(s.strip('()').replace('Mhz', ''))


def one_gate(n, qubit, g):
  ops = np.array([[workloads.NO_CTL, qubit]], dtype=np.int32)
  return ops, np.asarray(g, dtype=np.complex128).reshape(1, 4).view(np.float64).reshape(1, 8)


n = 30
work = {
    'stream only (T on index bit 0: one in-place sweep, empty op stream)': one_gate(n, n - 1, gates.tgate()),
    'one H on index bit 20 (one sweep, one register butterfly)': one_gate(n, n - 1 - 20, gates.hadamard()),
    'qft30 (3 sweeps)': workloads.qft_stream(range(n)).arrays(),
    'sup30 seed 0 (4 sweeps)': workloads.supremacy_stream(30, 20, seed=0).arrays(),
}
idle = smi()
print(f'idle: {idle[0]:.0f} W, sclk {idle[1]} MHz')
os.environ['QH_PLAN_CACHE'] = '1'
for name, (ops, g8) in work.items():
  with device.DeviceState(n, 128, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(5)
    for _ in range(8):
      st.run_stream(ops, g8); st.flush()
    st.sync()
    samples, stop = [], [False]

    def sampler():
      time.sleep(1.5)
      while not stop[0]:
        try:
          samples.append(smi())
        except Exception:  # pylint: disable=broad-except
          pass
        time.sleep(0.3)
    th = threading.Thread(target=sampler)
    th.start()
    st.reset_stats()
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < 7.0:
      for _ in range(20):
        st.run_stream(ops, g8); st.flush()
      st.sync()
      steps += 20
    dt = time.perf_counter() - t0
    stop[0] = True
    th.join()
    s = st.stats()
  sweeps = s['sweeps'] / steps
  ms_sweep = dt / s['sweeps'] * 1e3
  pw = sorted(x[0] for x in samples)[len(samples) // 2] if samples else float('nan')
  ck = sorted(x[1] for x in samples)[len(samples) // 2] if samples else 0
  print(f'{name}: {sweeps:.0f} sweeps per step, {ms_sweep:.3f} ms per sweep ({2 * 16 * 2**30 / ms_sweep / 1e6:.0f} GB/s), socket {pw:.0f} W '
        f'(median of {len(samples)}), sclk {ck} MHz -> {pw * ms_sweep * 1e-3:.2f} J per sweep ({(pw - idle[0]) * ms_sweep * 1e-3:.2f} J above idle)')
