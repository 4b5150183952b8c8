#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes to HBM bytes per launch.

Usage: collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced
streaming read, so it is doubled.  Both corrections are re-checked on known byte
counts in the same run: k_norm2 reads exactly S bytes, the rocclr fill kernel
writes exactly S bytes (reported under "calibration")."""
import collections
import csv
import json
import sys


def agg(path):
  d = collections.defaultdict(list)
  for r in csv.DictReader(open(path)):
    d[r['Kernel_Name']].append(float(r['Counter_Value']))
  return d


def main():
  fetch, write, out = sys.argv[1:4]
  f, w = agg(fetch), agg(write)
  res = {'unit': 'bytes per launch', 'formula': '(2*FETCH_SIZE + WRITE_SIZE) * 1024', 'kernels': {}, 'calibration': {}}
  for k in sorted(set(f) | set(w)):
    fv = sorted(f.get(k, [0.0]))
    wv = sorted(w.get(k, [0.0]))
    fm, wm = fv[len(fv) // 2], wv[len(wv) // 2]
    short = k.split('(')[0].replace('void ', '')
    res['kernels'][short] = {'launches': len(f.get(k, [])), 'fetch_KiB_median': fm, 'write_KiB_median': wm,
                             'hbm_bytes': (2 * fm + wm) * 1024}
    if 'k_norm2' in k:
      res['calibration']['read_only_k_norm2_bytes_(2*FETCH*1024)'] = 2 * fm * 1024
    if 'fillBuffer' in k:
      res['calibration']['write_only_fill_bytes_(WRITE*1024)'] = max(wv) * 1024
  json.dump(res, open(out, 'w'), indent=1)
  print(json.dumps(res, indent=1))


if __name__ == '__main__':
  main()
