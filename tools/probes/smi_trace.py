#!/usr/bin/env python3
"""Power / shader-clock trace of the GPU while a workload loops (backs DESIGN 4.3's "the clock sits at
1.6-2.0 GHz under a sweep": until round 3 that was inferred from s_memtime / s_memrealtime only).

  python tools/probes/smi_trace.py <workload> <seconds> <out.csv>
Starts tools/run_workload.py <workload> in a loop, samples every ~50 ms: hwmon power / sclk / mclk from sysfs
when the amdgpu driver exposes them, and `rocm-smi --showpower --showclocks --json` every ~1 s beside them."""
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
work, secs, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]


def read(path):
  try:
    with open(path) as f:
      return f.read().strip()
  except OSError:
    return ''


def sysfs_sample():
  row = {}
  for hw in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
    for name in ('power1_average', 'power1_input', 'freq1_input', 'freq2_input', 'temp1_input', 'temp2_input'):
      v = read(os.path.join(hw, name))
      if v:
        row[name] = v
    break
  for dev in glob.glob('/sys/class/drm/card*/device'):
    for name in ('pp_dpm_sclk', 'pp_dpm_mclk', 'gpu_busy_percent', 'mem_busy_percent'):
      v = read(os.path.join(dev, name))
      if v:
        cur = [ln for ln in v.splitlines() if ln.endswith('*')]
        row[name] = cur[0].split(':')[1].strip(' *') if cur else v.replace('\n', '|')[:60]
    break
  return row


def smi_sample():
  try:
    r = subprocess.run(['rocm-smi', '-d', '0', '--showpower', '--showclocks', '--showuse', '--json'], capture_output=True, text=True, timeout=10)
    d = json.loads(r.stdout)
    card = next(iter(d.values()))
    return {k: v for k, v in card.items() if any(s in k.lower() for s in ('power', 'sclk', 'mclk', 'fclk', 'busy', 'use'))}
  except Exception as e:  # pylint: disable=broad-except
    return {'smi_error': str(e)[:80]}


idle = sysfs_sample()
idle_smi = smi_sample()
p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'run_workload.py'), work, '100000'], stdout=subprocess.DEVNULL,
                     stderr=subprocess.DEVNULL)
rows = []
t0 = time.time()
next_smi = t0 + 6.0          # (let the workload start: import + state allocation)
try:
  while time.time() - t0 < secs:
    r = sysfs_sample()
    r['t'] = round(time.time() - t0, 3)
    if time.time() >= next_smi:
      r.update({'smi_' + k: v for k, v in smi_sample().items()})
      next_smi = time.time() + 1.0
    rows.append(r)
    time.sleep(0.05)
finally:
  p.kill()
  p.wait()
keys = ['t'] + sorted({k for r in rows for k in r if k != 't'})
with open(out, 'w') as f:
  f.write('# workload %s looping; idle before start: %s ; rocm-smi idle: %s\n' % (work, json.dumps(idle), json.dumps(idle_smi)))
  f.write(','.join(keys) + '\n')
  for r in rows:
    f.write(','.join(str(r.get(k, '')).replace(',', ';') for k in keys) + '\n')
print('wrote', out, len(rows), 'samples; last:', rows[-1] if rows else None)
