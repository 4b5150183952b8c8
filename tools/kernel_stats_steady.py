#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace CSV, over the dispatches AFTER the warm-up only.

Usage: kernel_stats_steady.py <kernel_trace.csv> <skip_first_n_sweep_dispatches> <out.csv>
rocprofv3's own --stats averages every dispatch, cold first steps included (3 x mean > ms_per_step);
this prints count / mean / median / min / max in ms per kernel name with the first N k_sweep dispatches dropped."""
import collections
import csv
import statistics
import sys

path, skip, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.OrderedDict()
seen_sweeps = 0
for r in rows:
  name = r['Kernel_Name'].split('(')[0].replace('void ', '')
  if 'k_sweep' in name:
    seen_sweeps += 1
    if seen_sweeps <= skip:
      continue
  elif seen_sweeps <= skip:
    continue
  by.setdefault(name, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
with open(out, 'w') as f:
  f.write(f'# {path}: dispatches after the first {skip} k_sweep launches (warm-up steps dropped); durations in ms\n')
  f.write('kernel,count,mean_ms,median_ms,min_ms,max_ms,total_ms\n')
  for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    f.write(f'{name},{len(v)},{statistics.mean(v):.4f},{statistics.median(v):.4f},{min(v):.4f},{max(v):.4f},{sum(v):.3f}\n')
print(open(out).read())
