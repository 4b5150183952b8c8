#!/usr/bin/env python3
"""ONE summary JSON of a tools/scale_ladder.sh run: per rung the measured ms / value beside the prediction of DESIGN 8
(qcc_amd.sharded.predict_step_ms), weak and strong scaling efficiency against the N = 1 rung, and the correctness evidence each
multi-rank line carries (parity_max_abs, rccl_ranks, exchange_verified).  usage: scale_summary.py OUTDIR > summary.json"""
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcc_amd import sharded  # noqa: E402


def load(path):
  for ln in open(path):
    if ln.startswith('{'):
      return json.loads(ln)
  return None


def main():
  out_dir = sys.argv[1]
  rungs = {}
  for f in sorted(glob.glob(os.path.join(out_dir, '*.json'))):
    m = re.match(r'(weak|strong)_q(\d+)_n(\d+)\.json', os.path.basename(f))
    if not m:
      continue
    kind, q, n = m.group(1), int(m.group(2)), int(m.group(3))
    d = load(f)
    pred = sharded.predict_step_ms(q, n)
    row = {'qubits': q, 'gpus': n, 'predicted_ms': {'expected': pred['expected_ms'], 'best_case': pred['best_case_ms']}}
    if d is None:
      row['error'] = 'no JSON line'
    elif d.get('error'):
      row.update(error=d['error'], error_stage=d.get('error_stage'))
    else:
      row.update(ms_per_step=d['ms_per_step'], value=d['value'], parity_max_abs=d.get('parity_max_abs'), rccl_ranks=d.get('rccl_ranks'),
                 exchange_verified=d.get('exchange_verified'), exchange_path=d.get('exchange_path'),
                 xgmi_GBps_per_rank=d.get('xgmi_GBps_per_rank'), exchange_ms_per_step_rank0=d.get('exchange_ms_per_step_rank0'),
                 sweeps_per_step=(d.get('predicted_ms_per_step') or {}).get('sweeps'),
                 measured_over_expected=d['ms_per_step'] / pred['expected_ms'])
    rungs.setdefault(kind, []).append(row)
  summary = {'ladder': rungs}
  for kind in ('weak', 'strong'):
    rows = sorted(rungs.get(kind, []), key=lambda r: r['gpus'])
    base = next((r for r in rows if r['gpus'] == 1 and 'value' in r), None)
    if base is None:
      continue
    for r in rows:
      if 'value' not in r:
        continue
      if kind == 'weak':       # per-GPU work fixed: value is whole-job throughput in 2^30-amplitude gate applications
        r['efficiency'] = r['value'] / (r['gpus'] * base['value'])
      else:                    # total work fixed: speed-up over N GPUs
        r['efficiency'] = base['ms_per_step'] / r['ms_per_step'] / r['gpus']
  ok = [r for rows in rungs.values() for r in rows if r['gpus'] > 1 and 'value' in r]
  summary['all_multi_rank_lines_verified'] = bool(ok) and all(
      r.get('exchange_verified') and r.get('rccl_ranks') == r['gpus'] and (r.get('parity_max_abs') or 1) <= 1e-10 for r in ok)
  print(json.dumps(summary, indent=1))


if __name__ == '__main__':
  main()
