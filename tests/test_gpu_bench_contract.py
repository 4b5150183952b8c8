"""bench.py end to end on the GPU at a small size: one JSON line with the contract's keys, the
roofline and cpu_baseline objects, and the sharded code path with a world of one."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
  e = dict(os.environ)
  e.update(env or {})
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], capture_output=True, text=True, timeout=900,
                     cwd=ROOT, env=e)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
  assert len(lines) == 1, r.stdout[-1000:]
  return json.loads(lines[0])


def test_bench_line_small():
  d = _run('--qubits', '24', '--steps', '3', '--warmup', '2', '--cpu-qubits', '24', '--cpu-gates', '6')
  for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
    assert k in d, k
  assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 2 and d['dtype'] == 'f64' and d['vs_baseline'] is None
  assert d['config']['qubits'] == 24 and 'workload' in d['config'] and 'model' not in d['config']
  r = d['roofline']
  assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and r['unit'] == 'GB/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
  assert r['bytes_per_launch'] == 2 * 16 * 2 ** 24
  c = d['cpu_baseline']
  assert c['cores'] == 1 and c['kind'] in ('reference', 'port') and c['value'] > 0 and len(c['binary_sha16']) == 16
  assert c['all_cores']['cores'] >= 1 and c['all_cores']['value'] > 0
  assert abs(d['norm2'] - 1) < 1e-10
  # value = gates x steps x (2^n / 2^30) / wall
  assert abs(d['value'] - 300 * 3 * 2 ** (24 - 30) / (d['ms_per_step'] * 3e-3)) / d['value'] < 1e-6
  assert 'cached_plan' in d and d['cached_plan']['ms_per_step'] > 0
  # no GPU-over-CPU ratio at the top level of the line; the profile's traffic figure is named as such
  assert 'gpu_over_cpu' not in d and c['gpu_over_this_baseline'] > 1
  assert 'traffic_from_profile' in r and set(r['traffic_from_profile']) == {'bytes', 'source'}


def test_bench_headline_extras_at_full_size():
  """The parts of the default line that only exist at the headline's size (30 qubits): `roofline.traffic` measured in the run
  (rocprofv3 PMC passes in a child process -- or the committed profile's figure with the reason, where rocprofv3 cannot run) and
  `roofline.energy` (Joules per k_sweep launch: stream-only floor vs the QFT).  Neither may cost the line."""
  d = _run('--steps', '3', '--warmup', '1', '--no-configs', '--no-ladder-base', '--no-cpu-baseline', '--no-cached-plan')
  r = d['roofline']
  assert d['config']['qubits'] == 30 and r['bytes_per_launch'] == 2 * 16 * 2 ** 30 and 0.5 < r['frac'] < 1.0
  assert r['traffic'] is not None and abs(r['traffic'] / r['bytes_per_launch'] - 1) < 0.01          # no wasted HBM traffic
  assert 'measured in this run' in r['traffic_source'] or 'committed profile' in r['traffic_source']
  assert r['traffic_from_profile']['bytes'] is not None
  e = r['energy']
  if e is not None and 'error' not in e:                        # (rocm-smi on the box)
    assert 0.8 < e['frac_of_floor_time'] <= 1.05 and 0.3 < e['frac_of_floor_energy'] <= 1.05
    assert e['stream_only']['J_per_launch'] > 1 and e['qft30']['socket_W'] > e['stream_only']['socket_W'] > e['idle_W'] * 0.5


def test_bench_sharded_world_of_one():
  d = _run('--sharded', '--qubits', '22', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', env={'QCC_EXCHANGE': 'native'})
  assert d['n_gpus'] == 1 and d['exchange_path'] == 'rccl' and d['exchanges_per_step'] == 0
  assert abs(d['norm2'] - 1) < 1e-10


@pytest.mark.parametrize('ranks', [2, 4])
def test_bench_multi_rank_launch_on_one_gpu(ranks):
  """The driver's multi-GPU command line (python -m torch.distributed.run ... bench.py --gpus N), with the ranks sharing
  the one GPU of the test box (gloo between them, the engine's exchange over the host-staged transport): the barrier /
  MAX-over-ranks timing, one JSON line from rank 0, the exchange accounting, the norm."""
  import socket
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  e = dict(os.environ)
  e.update(QCC_DIST_BACKEND='gloo', QCC_PRELOAD_TORCH='1')
  r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={ranks}', '--master-addr', '127.0.0.1',
                      '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(ranks), '--qubits', '24', '--steps', '2',
                      '--warmup', '1'], capture_output=True, text=True, timeout=900, cwd=ROOT, env=e)
  assert r.returncode == 0, r.stderr[-3000:]
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
  assert len(lines) == 1, r.stdout[-1000:]
  d = json.loads(lines[0])
  assert d['n_gpus'] == ranks and d['steps'] == 2 and d['scaling'] == 'weak' and d['exchange_path'] == 'host-staged'
  assert d['exchanges_per_step'] >= 1 and d['xgmi_bytes_per_rank_per_step'] > 0
  assert abs(d['norm2'] - 1) < 1e-10
  assert abs(d['value'] - 300 * 2 * 2 ** (24 - 30) / (d['ms_per_step'] * 2e-3)) / d['value'] < 1e-6
  # correctness evidence inside the line: sampled amplitudes of one more QFT vs the closed form on every rank,
  # what the communicator says about itself, and that the ranks compared their exchange geometry before data moved
  assert d['parity_max_abs'] is not None and d['parity_max_abs'] < 1e-10 and d['parity_samples_per_rank'] == 64
  assert d['rccl_ranks'] == ranks and d['exchange_verified'] is True and d['exchange_geometry_checks'] >= 1
  # the prediction the measurement is judged against, and the per-rank memory plan checked before allocating
  pr = d['predicted_ms_per_step']
  assert pr['expected_ms'] >= pr['best_case_ms'] > 0 and pr['exchanges'] == d['exchanges_per_step'] and pr['sweeps'] >= 1
  mp = d['memory_plan']
  assert mp['ranks'] == ranks and mp['shard_bytes'] == 16 << (24 - ranks.bit_length() + 1) and mp['fits_in_place'] is True
  assert mp['staging_bytes'] > 0 and mp['free_bytes'] > mp['need_relayout_bytes']
  assert 'memory plan: shard' in r.stderr
