#!/usr/bin/env python3
"""bench.py -- headline benchmark: n-qubit QFT gate application on MI355X.

Metric (BASELINE.json): gate-applies/sec + achieved HBM GB/s, 30-qubit QFT,
complex128, one GPU.  A "step" is one pass of the hot path over one batch: the
full 465-gate stream of qc.qft on 30 qubits (src/lib/circuit.py:320-328), every
gate submitted through the C-ABI (qh_apply1 / qh_applyc), state resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  N = 1: BASELINE config 2, the 30-qubit QFT (the headline and the roofline kernel); the line also
         carries `ladder_base` = the 33-qubit QFT on the same GPU, the N=1 point of config 5's ladder,
         `configs` = the other single-GPU BASELINE configurations timed the same way (config 3 supremacy-30,
         config 4 Grover-34, and the QFT at the reference's default width complex64), each with its own
         roofline, and `single_shot_ms` = ONE qft on a cold queue, planning included, followed by a read.
  N > 1: launched by torch.distributed.run, one rank per GPU; BASELINE config 5's ladder --
         34 / 35 / 36 qubits on 2 / 4 / 8 GPUs, the state sharded by its top log2(N) index bits,
         2^33 amplitudes = 128 GiB per GPU (weak scaling).  --qubits overrides.
  `value` is in units of 2^30-amplitude gate applications in every case, so the lines compare.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def default_qubits(world):
  """BASELINE.json: config 2 (30 qubits) on one GPU; config 5's ladder on several -- 34 / 35 / 36 qubits on
  2 / 4 / 8 GPUs, 2^33 amplitudes = 128 GiB per GPU."""
  g = int(math.log2(world))
  assert 1 << g == world, 'number of GPUs must be a power of two'
  return 30 if world == 1 else 33 + g


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=3)   # (the second step after a cold start still runs 5% slow)
  ap.add_argument('--qubits', type=int, default=0,
                  help='default: 30 on one GPU (BASELINE config 2); 33 + log2(gpus) on several (config 5 ladder)')
  ap.add_argument('--no-ladder-base', action='store_true', help='N=1: skip the extra 33-qubit measurement')
  ap.add_argument('--no-configs', action='store_true', help='N=1: skip configs 3, 4, complex64 and the single-shot timing')
  ap.add_argument('--fusion', type=int, default=-1, help='0 per-gate kernels, 1 fused sweeps (default)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-cached-plan', action='store_true', help='skip the extra steps timed with the plan cache on')
  ap.add_argument('--no-energy', action='store_true', help='N=1: skip the Joules-per-launch measurement (two 4-s loops under rocm-smi)')
  ap.add_argument('--no-live-traffic', action='store_true',
                  help='N=1: do not measure roofline.traffic with rocprofv3 in this run (the committed profile is quoted instead)')
  ap.add_argument('--sharded', action='store_true', help='use the multi-GPU layer even with one rank (smoke)')
  ap.add_argument('--cpu-qubits', type=int, default=30)
  ap.add_argument('--cpu-gates', type=int, default=24, help='gates of the stream timed on the CPU (24 of 465: ~21 s on one thread)')
  return ap.parse_args()


def class_pass(st, ops, g8, sel, reps):
  """Replay only the gates selected by `sel`, bracketed by HIP events on the
  engine's stream; returns (avg ms per kernel launch, launches, alg bytes/launch)."""
  st.sync()
  st.reset_stats()
  st.timer_begin()
  for _ in range(reps):
    st.run_stream(ops[sel], g8[sel])
  ms = st.timer_end()
  s = st.stats()
  launches = max(1, s['kernels_launched'])
  return ms / launches, launches // reps, s['bytes_swept'] / launches, s['bytes_algorithmic'] / launches


def unfused_classes(eng, ops, g8, n):
  """The per-gate kernels by class, each class replayed alone between HIP events: (avg ms per launch, launches, algorithmic
  bytes per launch, bytes moved per launch).  A CU1 whose control or target lies INSIDE the 128-byte line (index bits
  0..2: 8 complex128 amplitudes) changes 16 / 32 / 64 bytes of every line it visits and has to move whole lines
  (kernels_gate.hip.h:10-17; L2 / HBM work in 128-byte lines): its traffic is 2x its algorithmic bytes by construction,
  so SURVEY 8(d)'s 0.70 on ALGORITHMIC bytes is out of reach for that class; it is shown apart from the CU1s on bits >= 3."""
  from qcc_amd import workloads
  is_ctl = ops[:, 0] != workloads.NO_CTL
  low = is_ctl & (((n - 1 - ops[:, 0]) < 3) | ((n - 1 - ops[:, 1]) < 3))
  ms_d, n_d, swept_d, alg_d = class_pass(eng, ops, g8, is_ctl & ~low, 1)
  ms_l2, n_l2, swept_l2, alg_l2 = class_pass(eng, ops, g8, low, 1)
  ms_p, n_p, swept_p, alg_p = class_pass(eng, ops, g8, ~is_ctl, 1)
  return {'k_diag (CU1 on index bits >= 3, S/2 per launch)': (ms_d, n_d, alg_d, swept_d),
          'k_diag (CU1 touching index bits 0..2, inside the 128-byte line: whole lines rewritten)': (ms_l2, n_l2, alg_l2, swept_l2),
          'k_pair (H, 2S per launch)': (ms_p, n_p, alg_p, swept_p)}


def pmc_traffic(kernel_substr, fused, name=None):
  """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
  passes (profiles/rNN/traffic_*.json, produced by tools/collect_traffic.py on
  this same bench command / tools/run_workload.py for the other configs).  None when no pass has been committed."""
  import glob
  name = name or ('traffic_fused.json' if fused else 'traffic_unfused.json')
  files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', name)))
  if not files:
    return None, None
  d = json.load(open(files[-1]))
  best = None
  for k, v in d['kernels'].items():
    if kernel_substr in k and (best is None or v['launches'] > best['launches']):
      best = v
  if best is None:
    return None, None
  return best['hbm_bytes'], os.path.relpath(files[-1], ROOT)


def live_traffic(workload='qft30', reps=2, timeout_s=150):
  """HBM bytes per k_sweep launch MEASURED IN THIS RUN: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; with
  --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes: counters in KiB, FETCH_SIZE doubled on gfx950 for
  16-B-per-lane streaming reads) over tools/run_workload.py <workload> in a child process.  Returns (bytes, source) or
  (None, why) -- no rocprofv3 on the box, already running under a profiler, a pass that fails or times out: the caller then
  falls back to the committed profile and says so."""
  import csv
  import glob
  import shutil
  import subprocess
  import tempfile
  exe = shutil.which('rocprofv3')
  if exe is None:
    return None, 'rocprofv3 not on PATH'
  if any(k.startswith(('ROCPROFILER_', 'ROCPROF_', 'ROCP_TOOL')) for k in os.environ):
    return None, 'already running under a profiler'
  med = {}
  tmp = tempfile.mkdtemp(prefix='qcc_pmc_', dir='/tmp')
  try:
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
      d = os.path.join(tmp, ctr)
      env = dict(os.environ, TMPDIR='/tmp')
      r = subprocess.run([exe, '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', d, '-o', 'p', '--', sys.executable,
                          os.path.join(ROOT, 'tools', 'run_workload.py'), workload, str(reps)], cwd='/tmp', env=env,
                         capture_output=True, text=True, timeout=timeout_s)
      files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
      if r.returncode != 0 or not files:
        return None, f'rocprofv3 --pmc {ctr} failed (rc {r.returncode})'
      vals = [float(row['Counter_Value']) for row in csv.DictReader(open(files[0]))
              if 'k_sweep' in row['Kernel_Name'] and row['Counter_Name'] == ctr]
      if not vals:
        return None, f'no k_sweep dispatch in the {ctr} pass'
      vals.sort()
      med[ctr] = vals[len(vals) // 2]
    return (2 * med['FETCH_SIZE'] + med['WRITE_SIZE']) * 1024, ('measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE '
                                                               f'(separate passes) over tools/run_workload.py {workload} {reps}; '
                                                               '(2 x FETCH_SIZE + WRITE_SIZE) x 1024, median over the k_sweep dispatches')
  except Exception as e:  # pylint: disable=broad-except
    return None, f'{type(e).__name__}: {e}'
  finally:
    shutil.rmtree(tmp, ignore_errors=True)


def energy_roofline(device_index, seconds=4.0):
  """Joules per k_sweep launch -- the second roofline of these sweeps (the socket's 1 400 W limit).  Two loops of
  `seconds` each on a 30-qubit state while rocm-smi is sampled (~3 Hz, median): (a) a sweep with an EMPTY op stream (one T gate: the
  tile stream of k_sweep and nothing else) = the floor in time AND energy of this launch shape, (b) the headline's QFT.  J per launch
  = socket W x ms per launch.  None where rocm-smi is not available."""
  import shutil
  import subprocess
  import threading
  from qcc_amd import device, gates, native, workloads
  if shutil.which('rocm-smi') is None:
    return None

  def smi():
    r = subprocess.run(['rocm-smi', '-d', str(device_index), '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10)
    card = next(iter(json.loads(r.stdout).values()))
    return float(card['Current Socket Graphics Package Power (W)']), int(card.get('sclk clock speed:', '(0Mhz)').strip('()').replace('Mhz', ''))

  try:
    idle_w, _ = smi()
    n = 30
    t_ops = np.array([[workloads.NO_CTL, n - 1]], dtype=np.int32)
    t_g = np.asarray(gates.tgate(), dtype=np.complex128).reshape(1, 4).view(np.float64).reshape(1, 8)
    out = {'idle_W': idle_w, 'method': f'rocm-smi socket power, median over a {seconds:g}-s loop per workload; J = W x ms per k_sweep launch'}
    for key, (ops, g8) in (('stream_only', (t_ops, t_g)), ('qft30', workloads.qft_stream(range(n)).arrays())):
      with device.DeviceState(n, 128, device=device_index, fusion=native.QH_FUSE_SWEEP) as st:
        st.init_basis(5)
        for _ in range(6):
          st.run_stream(ops, g8)
          st.flush()
        st.sync()
        samples, stop = [], [False]

        def sampler():
          time.sleep(1.0)
          while not stop[0]:
            try:
              samples.append(smi())
            except Exception:  # pylint: disable=broad-except
              pass
            time.sleep(0.25)
        th = threading.Thread(target=sampler)
        th.start()
        st.reset_stats()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
          for _ in range(20):
            st.run_stream(ops, g8)
            st.flush()
          st.sync()
        dt = time.perf_counter() - t0
        stop[0] = True
        th.join()
        launches = max(1, st.stats()['kernels_launched'])
      if not samples:
        return None
      w = sorted(x[0] for x in samples)[len(samples) // 2]
      ck = sorted(x[1] for x in samples)[len(samples) // 2]
      ms = dt / launches * 1e3
      out[key] = {'ms_per_launch': ms, 'socket_W': w, 'sclk_MHz': ck, 'J_per_launch': w * ms * 1e-3, 'samples': len(samples)}
    out['frac_of_floor_energy'] = out['stream_only']['J_per_launch'] / out['qft30']['J_per_launch']
    out['frac_of_floor_time'] = out['stream_only']['ms_per_launch'] / out['qft30']['ms_per_launch']
    out['note'] = ('stream_only = k_sweep with an empty op stream (same launch shape, same bytes): what the memory system alone takes and draws; '
                   'the distance of qft30 from it is the op stream (VALU 4.0 G + scalar 2.1 G instructions per QFT), paid in Joules under the 1 400 W limit')
    return out
  except Exception as e:  # pylint: disable=broad-except
    return {'error': repr(e)}


def cpu_baseline(args, ops, g8):
  """Reference xgates.cc build (oracle/_ref) if it travelled, else our C port,
  single thread, on a bounded sample of the same stream."""
  from tests import oracle_lib
  n = args.cpu_qubits
  psi = None
  while psi is None and n >= 24:
    try:
      psi = np.zeros(1 << n, dtype=np.complex128)
      psi[0x2CB9A5E3 & ((1 << n) - 1)] = 1.0
    except MemoryError:
      n -= 2
  full = len(ops)
  # same stream shape at n qubits
  from qcc_amd import workloads
  o, g = workloads.qft_stream(range(n)).arrays()
  k = max(1, args.cpu_gates)
  pick = np.unique(np.linspace(0, len(o) - 1, k).astype(int))
  xg = oracle_lib.load_ref_xgates()
  kind = 'reference' if xg is not None else 'port'
  gc = np.ascontiguousarray(g).view(np.complex128).reshape(-1, 4)
  # warm the pages with one H so that first-touch cost is not billed to the gate
  if xg is not None:
    xg.apply1(psi, gc[0], n, n - 1, 128)
  orc = oracle_lib.load(fast=True) if xg is None else None
  if orc is not None:
    orc.apply1(psi, gc[0], n, n - 1)
  t0 = time.perf_counter()
  for i in pick:
    c, t = int(o[i, 0]), int(o[i, 1])
    if xg is not None:
      if c == workloads.NO_CTL:
        xg.apply1(psi, gc[i], n, t, 128)
      else:
        xg.applyc(psi, gc[i], n, c, t, 128)
    else:
      if c == workloads.NO_CTL:
        orc.apply1(psi, gc[i], n, t)
      else:
        orc.applyc(psi, gc[i], n, c, t)
  dt = time.perf_counter() - t0
  rate = len(pick) / dt
  # scale to the benchmark's state size: cost per gate is linear in 2^n
  scale = 2.0 ** (n - args.qubits_shard)
  # all host cores, for context: our OpenMP restatement of the same loops
  # (the reference has no threaded path), same sampled gates
  all_cores = None
  try:
    omp = oracle_lib.load(omp=True)
    del psi
    psi2 = np.empty(1 << n, dtype=np.complex128)                   # untouched pages ...
    omp.init_basis_mt(psi2, n, 0x2CB9A5E3 & ((1 << n) - 1))          # ... first touched by the threads that use them (NUMA)
    omp.run_stream_mt(psi2, n, o[pick[:1]], g[pick[:1]])          # thread pool warm-up
    t1 = time.perf_counter()
    omp.run_stream_mt(psi2, n, o[pick], g[pick])
    dt_mt = time.perf_counter() - t1
    del psi2
    all_cores = {'value': len(pick) / dt_mt * scale, 'unit': 'gate-applies/s', 'cores': os.cpu_count(), 'kind': 'port',
                 'sample': f'the same {len(pick)} gates, oracle/xgates_oracle.c oracle_run_stream_c128_mt, OpenMP, {dt_mt:.2f} s'}
  except Exception as e:  # pylint: disable=broad-except
    all_cores = {'error': str(e)}
  import hashlib
  bin_path = (os.path.join(ROOT, 'oracle', '_ref', 'libxgates.so') if kind == 'reference'
              else os.path.join(ROOT, 'oracle', '_build', 'liboracle_fast.so'))
  with open(bin_path, 'rb') as f:
    bin_hash = hashlib.sha256(f.read()).hexdigest()[:16]
  return {
      'value': rate * scale, 'unit': 'gate-applies/s', 'cores': 1, 'kind': kind,
      'binary': os.path.relpath(bin_path, ROOT), 'binary_sha16': bin_hash, 'all_cores': all_cores,
      'sample': (f'{len(pick)} gates (indices {pick.tolist()}) of the {len(o)}-gate {n}-qubit QFT stream, '
                 f'complex128, single thread, {dt:.1f} s'
                 + ('' if n == args.qubits_shard else f'; scaled x{scale:g} to 2^{args.qubits_shard} amplitudes')
                 + (' ; reference src/lib/xgates.cc built -O3 -ffast-math by oracle/Makefile'
                    if kind == 'reference' else ' ; oracle/xgates_oracle.c -O3 -march=native')),
      'host_cpus': os.cpu_count(),
  }


def timed_steps(eng, ops, g8, steps, warmup, dist):
  """W untimed steps, then exactly K steps bracketed by barrier + device sync on both sides.
  Returns (wall seconds: max over ranks, HIP-event ms on the engine's stream, engine stats)."""
  for _ in range(warmup):
    eng.run_stream(ops, g8)
    eng.flush()  # a step is one observable qc.qft(): never fuse across steps
  eng.sync()
  eng.reset_stats()
  if dist is not None:
    import torch
    torch.cuda.synchronize()
    dist.barrier()
  t0 = time.perf_counter()
  eng.timer_begin()
  eng.timer_lap()
  for _ in range(steps):
    eng.run_stream(ops, g8)
    eng.flush()
    eng.timer_lap()          # an event on the stream per step (no host wait): the per-step times behind `median`
  ev_ms = eng.timer_end()  # flushes + waits for the stream
  eng.sync()
  if dist is not None:
    import torch
    torch.cuda.synchronize()
    dist.barrier()
  wall = time.perf_counter() - t0
  if dist is not None:
    import torch
    t = torch.tensor([wall], dtype=torch.float64, device=eng._red_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
  st = eng.stats()
  st['step_ms'] = eng.timer_laps()
  return wall, ev_ms, st


def sharded_parity(eng, n, x, ops, g8, dist, samples=64):
  """Correctness evidence INSIDE the multi-GPU line (the first run on several GPUs is a test, not only a number): the
  state is re-initialised, ONE QFT runs through the same path as the timed steps -- sweeps, exchange over the links,
  sweeps -- and every rank compares `samples` of the amplitudes it holds with the closed form of the QFT of a basis
  state (workloads.qft_analytic: exp(2 pi i bitrev(x) k / N) / sqrt(N)); the largest error over all ranks comes back.
  A misplaced block keeps the norm and fails this."""
  import torch
  from qcc_amd import workloads
  eng.init_basis(x)
  eng.run_stream(ops, g8)
  eng.flush()
  eng.sync()
  rng = np.random.default_rng(1234 + eng.rank)
  nloc = eng.nloc
  # physical indices inside this rank's shard, spread over its blocks (the blocks another rank sent are the interesting ones)
  local = rng.integers(0, 1 << nloc, size=samples, dtype=np.uint64)
  local[:8] = [0, (1 << nloc) - 1, 1, 1 << (nloc - 1), (1 << (nloc - 1)) - 1, 1 << (nloc - 2), 3 << (nloc - 2), 5][:8]
  logical = [eng.phys_to_logical((eng.rank << nloc) | int(v)) for v in local]
  got = np.array([eng.amplitude_local(k) for k in logical], dtype=np.complex128)
  want = workloads.qft_analytic(n, x, np.array(logical, dtype=np.uint64))
  err = float(np.max(np.abs(got - want)))
  t = torch.tensor([err], dtype=torch.float64, device=eng._red_device())
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


def ladder_base(device_index, fusion, steps=3):
  """The N=1 point of BASELINE config 5's ladder (33/34/35/36 qubits on 1/2/4/8 GPUs, 2^33
  amplitudes = 128 GiB per GPU): a 33-qubit QFT on this GPU, outside the headline's timed region."""
  from qcc_amd import device, native, workloads
  n = 33
  try:
    eng = device.DeviceState(n, 128, device=device_index, fusion=fusion)
  except native.QhError as e:
    return {'qubits': n, 'error': str(e)}
  with eng:
    ops, g8 = workloads.qft_stream(range(n)).arrays()
    eng.init_basis(0x12CB9A5E3 & ((1 << n) - 1))
    wall, ev_ms, st = timed_steps(eng, ops, g8, steps, 1, None)
    norm2 = eng.norm2()
  units = 2 ** (n - 30)
  return {'qubits': n, 'gates_per_step': len(ops), 'steps': steps, 'ms_per_step': wall / steps * 1e3,
          'median_ms_per_step': float(np.median(st['step_ms'])) if st['step_ms'] else None,
          'value': len(ops) * steps * units / wall, 'unit': 'gate-applies/s (2^30-amplitude units)',
          'kernels_per_step': st['kernels_launched'] / steps,
          'hbm_GBps_swept': st['bytes_swept'] / (ev_ms * 1e-3) / 1e9,
          'roofline_frac': st['bytes_swept'] / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 'norm2': norm2}


def config_line(name, n, bw, ops, g8, init, device_index, steps, warmup, note, traffic_file=None):
  """One of the other single-GPU BASELINE configurations, timed like the headline: W warm-up steps, K timed steps
  (wall clock between device syncs; HIP events per step), planned from scratch every step.  Its own roofline:
  bytes per k_sweep launch from the plans (engine stats), average launch duration from the HIP events."""
  from qcc_amd import device, native
  try:
    eng = device.DeviceState(n, bw, device=device_index, fusion=native.QH_FUSE_SWEEP)
  except native.QhError as e:
    return {'workload': name, 'qubits': n, 'skipped': str(e)}
  try:
    with eng:
      eng.init_basis(init)
      wall, ev_ms, st = timed_steps(eng, ops, g8, steps, warmup, None)
      norm2 = eng.norm2()
  except native.QhError as e:         # (one entry's failure must not cost the line)
    return {'workload': name, 'qubits': n, 'error': str(e)}
  launches = max(1, st['kernels_launched'])
  bytes_l = st['bytes_swept'] / launches
  ms_region = ev_ms / launches
  # The kernel's launch duration: the MEDIAN step's HIP-event time / launches per step.  The mean over the timed region also holds the
  # idle stretch in front of the first timed step's launches (its planning has no GPU work to hide behind: ~18 ms for a circuit the
  # planner searches, first_step_ms), which is host time, not kernel time -- rocprofv3's per-dispatch mean (profiles/) agrees with
  # the median-based figure; the region's mean stays in the line beside it.
  ms_l = float(np.median(st['step_ms'])) / (launches / steps) if st['step_ms'] else ms_region
  traffic, tsrc = pmc_traffic('k_sweep', True, traffic_file) if traffic_file else (None, None)
  return {'workload': name, 'qubits': n, 'dtype': 'f64' if bw == 128 else 'f32', 'gates_per_step': len(ops), 'steps': steps,
          'warmup': warmup, 'ms_per_step': wall / steps * 1e3,
          'median_ms_per_step': float(np.median(st['step_ms'])) if st['step_ms'] else None,
          'step_ms_min_max': [float(min(st['step_ms'])), float(max(st['step_ms']))] if st['step_ms'] else None,
          'first_step_ms': float(st['step_ms'][0]) if st['step_ms'] else None,      # (planning of the first timed step has no GPU work to hide behind)
          'event_ms_per_step': ev_ms / steps, 'gate_applies_per_s': len(ops) * steps / wall,
          'sweeps_per_step': st['sweeps'] / steps,
          'effective_GBps_algorithmic': st['bytes_algorithmic'] / wall / 1e9,
          'roofline': {'bound': 'hbm', 'kernel': 'k_sweep', 'achieved': bytes_l / (ms_l * 1e-3) / 1e9, 'peak': HBM_PEAK_GBPS,
                       'unit': 'GB/s', 'frac': bytes_l / (ms_l * 1e-3) / 1e9 / HBM_PEAK_GBPS, 'avg_launch_ms': ms_l,
                       'avg_launch_ms_source': 'median step (HIP events) / launches per step', 'avg_launch_ms_over_timed_region': ms_region,
                       'bytes_per_launch': bytes_l, 'traffic': traffic, 'traffic_source': tsrc},
          'norm2': norm2, 'note': note}


def unfused_line(device_index):
  """The headline's gate stream through the ONE-GATE-PER-LAUNCH kernels (SURVEY 8(d) judges the 0.70 target on these too):
  2 timed steps of 465 launches, then each kernel class alone between HIP events."""
  from qcc_amd import device, native, workloads
  n = 30
  ops, g8 = workloads.qft_stream(range(n)).arrays()
  try:
    eng = device.DeviceState(n, 128, device=device_index, fusion=native.QH_FUSE_OFF)
  except native.QhError as e:
    return {'workload': 'unfused qft30', 'skipped': str(e)}
  with eng:
    eng.init_basis(0x12CB9A5E3 & ((1 << n) - 1))
    wall, ev_ms, st = timed_steps(eng, ops, g8, 2, 1, None)
    classes = unfused_classes(eng, ops, g8, n)
    norm2 = eng.norm2()
  dom = max(classes, key=lambda k: classes[k][0] * classes[k][1])
  return {'workload': '30-qubit QFT complex128 through the per-gate kernels (k_pair / k_diag), 465 launches per step', 'qubits': n,
          'dtype': 'f64', 'steps': 2, 'warmup': 1, 'ms_per_step': wall / 2 * 1e3, 'gate_applies_per_s': len(ops) * 2 / wall,
          'effective_GBps_algorithmic': st['bytes_algorithmic'] / wall / 1e9, 'hbm_GBps_moved': st['bytes_swept'] / wall / 1e9,
          'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': classes[dom][2] / classes[dom][0] / 1e6, 'peak': HBM_PEAK_GBPS,
                       'unit': 'GB/s', 'frac': classes[dom][2] / classes[dom][0] / 1e6 / HBM_PEAK_GBPS,
                       'traffic': pmc_traffic('k_diag', False)[0], 'traffic_source': pmc_traffic('k_diag', False)[1],
                       'classes': {k: {'avg_ms': v[0], 'launches_per_step': v[1], 'GBps_algorithmic': v[2] / v[0] / 1e6,
                                       'GBps_moved': v[3] / v[0] / 1e6, 'frac_algorithmic': v[2] / v[0] / 1e6 / HBM_PEAK_GBPS}
                                   for k, v in classes.items()}},
          'norm2': norm2}


def other_configs(device_index):
  from qcc_amd import workloads
  out = {}
  ops, g8 = workloads.supremacy_stream(30, 20, seed=0).arrays()
  out['config3_supremacy30_d20_seed0'] = config_line(
      '30-qubit supremacy.py random circuit, depth 20, random.seed(0) [BASELINE config 3]', 30, 128, ops, g8, 0, device_index, 20, 5,
      'op-heavy sweeps: bound by the socket power limit, not by HBM -- op streams alone 1 284 W at 2.39 GHz, with the tile streams 1 391 W of 1 400 W at 1.92 GHz (DESIGN 4.5, profiles/r04/smi_trace_sup30_*.csv)', 'traffic_sup30.json')
  for seed in (1, 2):          # SURVEY 8(d) config 3 names seeds 0, 1, 2
    ops, g8 = workloads.supremacy_stream(30, 20, seed=seed).arrays()
    out[f'config3_supremacy30_d20_seed{seed}'] = config_line(
        f'30-qubit supremacy.py random circuit, depth 20, random.seed({seed}) [BASELINE config 3, SURVEY 8(d) seeds 0-2]', 30, 128, ops, g8, 0,
        device_index, 20, 5, 'same family as seed 0; the number of sweeps depends on the instance (DESIGN 4.7 level search: 4 for seeds 0-4, the minimum under 13-bit tiles)')
  ops, g8 = workloads.qft_stream(range(30)).arrays()
  out['qft30_complex64'] = config_line(
      '30-qubit QFT at the reference\'s default width complex64 (src/lib/tensor.py:28)', 30, 64, ops, g8,
      0x12CB9A5E3 & ((1 << 30) - 1), device_index, 10, 3, 'state = 8 GiB; same gate stream as the headline', 'traffic_qft30c64.json')
  out['unfused_qft30'] = unfused_line(device_index)
  nb = 17
  ops, g8 = workloads.grover_stream(nb, [1, 0] * 8 + [1], iterations=1).arrays()
  out['config4_grover34_one_iteration'] = config_line(
      '34-qubit Grover (nbits 17), one iteration = oracle + diffusion, 256 GiB state [BASELINE config 4]', 2 * nb, 128, ops, g8,
      workloads.grover_initial_index(nb), device_index, 2, 1, 'in place: no room for a second buffer beside 256 GiB',
      'traffic_grover34.json')
  return out


def single_shot(device_index):
  """ONE 30-qubit QFT on an idle engine followed by a read: nothing hides the planner here.  (a) through the API
  mirror: circuit.qc().qft(reg) + maxprob(), Python gate construction included; (b) through the C-ABI: the gate
  stream submitted with one qh_apply_stream call, then qh_sync."""
  from qcc_amd import device, native, workloads
  from qcc_amd.lib import circuit, tensor
  out = {}
  try:
    n = 30
    tensor.set_tensor_width(128)
    os.environ['QH_PLAN_CACHE'] = '0'
    times = []
    for rep in range(3):
      qc = circuit.qc('single-shot')
      reg = qc.reg(n, 0x12CB9A5E3 & ((1 << n) - 1) if rep == 0 else rep)
      qc.flush()                               # the register is built on the device (qh_init_product: a product register stays
      qc.sync()                                # a list of factors until something needs the amplitudes), engine idle
      t0 = time.perf_counter()
      qc.qft(reg)
      bits, p = qc.maxprob()
      times.append((time.perf_counter() - t0) * 1e3)
      del qc
    out['qc_qft_then_maxprob_ms'] = {'runs': times, 'median': float(np.median(times))}
    ops, g8 = workloads.qft_stream(range(n)).arrays()
    times = []
    with device.DeviceState(n, 128, device=device_index, fusion=native.QH_FUSE_SWEEP) as st:
      for rep in range(4):
        st.init_basis(5 + rep)
        st.sync()
        t0 = time.perf_counter()
        st.run_stream(ops, g8)
        st.sync()
        times.append((time.perf_counter() - t0) * 1e3)
    out['c_abi_stream_then_sync_ms'] = {'runs': times, 'median': float(np.median(times[1:])),
                                        'note': 'first run allocates the second buffer and the op buffers'}
  except Exception as e:  # pylint: disable=broad-except
    out['error'] = repr(e)
  finally:
    tensor.set_tensor_width(None)
    # the circuits above parked their device states (two 16-GiB buffers each) for the next circuit of that shape:
    # the 128- and 256-GiB states that follow need the room
    from qcc_amd.lib import backend
    backend.drop_device_pool()
  return out


def main():
  args = parse()
  # every step plans its gate stream from scratch (the engine's plan cache would only save host
  # time that is hidden behind the previous step's kernels anyway; nothing is reused across steps)
  os.environ.setdefault('QH_PLAN_CACHE', '0')
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit('bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)')
  if world > 1 or args.sharded:
    os.environ['QCC_PRELOAD_TORCH'] = '1'  # one HIP runtime per process: torch's (qcc_amd/native.py)
  from qcc_amd import device, native, workloads
  if native.device_count() < 1:
    raise SystemExit('bench.py: no HIP device visible; the engine has no CPU fallback')
  if os.environ.get('QCC_DIST_BACKEND') == 'gloo':   # test mode: several ranks share the visible GPU(s)
    local_rank %= native.device_count()
  gbits = int(math.log2(world))
  assert 1 << gbits == world, 'number of GPUs must be a power of two'
  n = args.qubits or default_qubits(world)
  nloc = n - gbits
  args.qubits_shard = nloc
  fusion = args.fusion if args.fusion >= 0 else native.QH_FUSE_SWEEP

  def fail_line(stage, exc, eng=None):
    """Multi-GPU runs: a transport that cannot be set up, ranks that disagree about an exchange, a peer that never
    shows up (engine watchdog, QH_COMM_TIMEOUT_MS) end in ONE JSON line with "error" -- never in a silent switch to
    another data path, never in a hang.  Every failing rank reports on stderr; rank 0 (or the failing rank, if rank 0
    cannot know) prints the line; the process exits non-zero without waiting for peers that may be stuck."""
    msg = f'{type(exc).__name__}: {exc}'
    print(f'[bench.py rank {rank}/{world}] {stage} failed: {msg}', file=sys.stderr, flush=True)
    line = {'metric': 'gate-applies/sec (2^30-amplitude units), QFT', 'value': None, 'unit': 'gate-applies/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'{n}-qubit QFT complex128 (failed before a result)', 'qubits': n},
            'error': msg, 'error_stage': stage, 'error_rank': rank,
            'exchange_path': getattr(eng, 'exchange_path', None)}
    # ONE line on stdout: rank 0 prints it (errors of this kind are raised on every rank by design: TransportError, geometry
    # mismatch, watchdog); another rank prints only when it fails alone -- rank 0 is then stuck or gone and cannot speak.
    alone = bool(getattr(exc, 'local_only', False))
    if rank == 0 or alone:
      print(json.dumps(line), flush=True)
    try:
      if eng is not None:
        eng.eng.close() if hasattr(eng, 'eng') else eng.close()
    except Exception:  # pylint: disable=broad-except
      pass
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(3)          # (no destroy_process_group: peers may be stuck in a collective that will never complete)

  dist = None
  eng = None
  try:
    if world > 1 or args.sharded:
      from qcc_amd import sharded
      eng = sharded.ShardedState(n, fusion=fusion, local_rank=local_rank)
      dist = eng.dist
    else:
      eng = device.DeviceState(n, 128, device=local_rank, fusion=fusion)
  except Exception as e:  # pylint: disable=broad-except
    if world == 1:
      raise
    fail_line('setup (process group / engine / exchange transport)', e, eng)

  ops, g8 = workloads.qft_stream(range(n)).arrays()
  ngates = len(ops)
  x = 0x12CB9A5E3 & ((1 << n) - 1)
  try:
    eng.init_basis(x)
    wall, ev_ms, stats = timed_steps(eng, ops, g8, args.steps, args.warmup, dist)
  except Exception as e:  # pylint: disable=broad-except
    if world == 1:
      raise
    fail_line('timed steps', e, eng)

  # parity guard inside the bench: the norm (cheap, device-side) and -- with several ranks -- sampled amplitudes of one
  # more QFT against the closed form (a single-GPU run is covered by the GPU tests, which check every amplitude)
  norm2 = eng.norm2_global() if dist is not None else eng.norm2()
  parity = None
  if dist is not None:
    try:
      parity = sharded_parity(eng, n, x, ops, g8, dist)
    except Exception as e:  # pylint: disable=broad-except
      fail_line('parity check after the timed steps', e, eng)
    if not parity <= 1e-10:
      fail_line('parity check after the timed steps',
                RuntimeError(f'max |amplitude - closed form| over {world} ranks x 64 samples = {parity:.3e} > 1e-10 '
                             f'(norm2 {norm2:.15f}): amplitudes are misplaced or wrong'), eng)
  cached = None
  if (world == 1 and dist is None and fusion != native.QH_FUSE_OFF and os.environ.get('QH_PLAN_CACHE') == '0'
      and not args.no_cached_plan):
    # the same steps with the engine's plan cache on (loops over one circuit skip the planner): reported
    # beside the headline, which plans every step from scratch
    os.environ['QH_PLAN_CACHE'] = '1'
    w2, e2, s2 = timed_steps(eng, ops, g8, args.steps, 8, None)
    os.environ['QH_PLAN_CACHE'] = '0'
    cached = {'ms_per_step': w2 / args.steps * 1e3, 'event_ms_per_step': e2 / args.steps,
              'note': 'QH_PLAN_CACHE=1, 8 extra warm-up steps (the layouts of a relayout cycle each get their plan cached)'}

  out = None
  if rank == 0:
    steps = args.steps
    units = world * 2.0 ** (nloc - 30)  # one gate on a 2^n state = 2^(n-30) gate applications of 2^30 amplitudes
    value = ngates * steps * units / wall
    launches = max(1, stats['kernels_launched'])
    # ---- roofline of the dominant kernel --------------------------------------
    if world == 1 and dist is None:
      if fusion == native.QH_FUSE_OFF:
        classes = unfused_classes(eng, ops, g8, n)
        name = max(classes, key=lambda k: classes[k][0] * classes[k][1])
        ms_l, n_l, bytes_l = classes[name][:3]
        other = {k: {'avg_ms': v[0], 'launches_per_step': v[1], 'GBps_algorithmic': v[2] / v[0] / 1e6,
                     'GBps_moved': v[3] / v[0] / 1e6} for k, v in classes.items()}
      else:
        name = 'k_sweep (fused register-tile sweep, read S + write S per launch)'
        ms_l = ev_ms / launches
        bytes_l = stats['bytes_swept'] / launches
        other = {}
      achieved = bytes_l / (ms_l * 1e-3) / 1e9
      traffic, tsrc = pmc_traffic(name.split(' ')[0], fusion != native.QH_FUSE_OFF)
      traffic_profile = {'bytes': traffic, 'source': tsrc}
      roofline = {'bound': 'hbm', 'kernel': name, 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                  'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_source': tsrc,
                  'traffic_from_profile': traffic_profile,
                  'avg_launch_ms': ms_l, 'bytes_per_launch': bytes_l, 'classes': other}
    else:
      roofline = {'bound': 'hbm', 'achieved': stats['bytes_swept'] / (ev_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBPS,
                  'unit': 'GB/s', 'frac': stats['bytes_swept'] / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                  'traffic': None, 'note': 'per-GPU bytes swept / event time of rank 0 (includes exchange waits)'}
    out = {
        'metric': 'gate-applies/sec (2^30-amplitude units), QFT', 'value': value, 'unit': 'gate-applies/s',
        'n_gpus': world, 'steps': steps, 'warmup': args.warmup, 'ms_per_step': wall / steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': (f'{n}-qubit QFT complex128, {ngates} gates/step (qc.qft order), basis-state input, '
                                f'state resident in HBM, 2^{nloc} amplitudes per GPU'
                                + (' [BASELINE config 2]' if (world == 1 and n == 30) else '')
                                + (' [BASELINE config 5 ladder: 33/34/35/36 qubits on 1/2/4/8 GPUs]'
                                   if nloc == 33 else '')),
                   'qubits': n, 'gates_per_step': ngates, 'fusion': fusion,
                   'sharding': 'none' if world == 1 else f'top {gbits} index bits across {world} GPUs'},
        'whole_state_gate_applies_per_s': ngates * steps / wall,
        'effective_GBps_algorithmic': stats['bytes_algorithmic'] / wall / 1e9,
        'hbm_GBps_swept': stats['bytes_swept'] / wall / 1e9,
        'kernels_per_step': stats['kernels_launched'] / steps,
        'event_ms_per_step': ev_ms / steps, 'norm2': norm2,
        'median_ms_per_step': float(np.median(stats['step_ms'])) if stats.get('step_ms') else None,
        'step_ms_min_max': [float(min(stats['step_ms'])), float(max(stats['step_ms']))] if stats.get('step_ms') else None,
        'roofline': roofline,
    }
    if cached:
      out['cached_plan'] = cached
    if dist is not None:
      from qcc_amd import sharded as _sh
      # what this step SHOULD cost (sweeps at the measured single-GPU rate + the exchange over the links with the
      # overlap the slabs allow), from the sweep / exchange counts this run actually had -- beside the measurement
      out['predicted_ms_per_step'] = _sh.predict_step_ms(n, world, sweeps=stats['sweeps'] / steps,
                                                         exchanges=stats.get('exchanges', 0) / steps)
      out['memory_plan'] = getattr(eng, 'memory_plan', None)
      out['exchanges_per_step'] = stats.get('exchanges', 0) / steps
      out['xgmi_bytes_per_rank_per_step'] = stats.get('exchanged_bytes', 0) / steps
      out['exchange_ms_per_step_rank0'] = stats.get('exchange_seconds', 0.0) / steps * 1e3
      out['exchange_path'] = stats.get('exchange_path')
      out['exchange_geometry'] = stats.get('exchange_geometry')     # slabs, rounds, chunk, packed / direct, staging bytes (rank 0)
      out['relayout_on_every_rank'] = getattr(eng, 'relayout', None)
      out['parity_max_abs'] = parity                                   # vs the closed form, 64 amplitudes per rank, tolerance 1e-10
      out['parity_samples_per_rank'] = 64
      out['rccl_ranks'] = stats.get('comm_ranks_reported')             # what the communicator itself reports
      out['exchange_geometry_checks'] = stats.get('exchange_geometry_checks')
      out['exchange_verified'] = bool(world == 1 or (stats.get('exchange_geometry_checks') or 0) > 0)
      if stats.get('exchange_seconds', 0.0) > 0:
        out['xgmi_GBps_per_rank'] = stats.get('exchanged_bytes', 0) / stats['exchange_seconds'] / 1e9
  if dist is not None:
    eng.close()
    dist.destroy_process_group()
  else:
    eng.close()
  if rank == 0:
    extras = world == 1 and dist is None and n == 30 and fusion != native.QH_FUSE_OFF
    if extras and not args.no_live_traffic:
      # roofline.traffic measured in THIS run (VERDICT r05 "what's weak" 8): the engine is closed, a child process runs the
      # same 30-qubit QFT under rocprofv3's PMC passes; the committed profile's figure stays beside it as traffic_from_profile
      t_live, why = live_traffic('qft30')
      if t_live is not None:
        out['roofline']['traffic'], out['roofline']['traffic_source'] = t_live, why
      else:
        out['roofline']['traffic_source'] = f"{out['roofline']['traffic_source']} (committed profile; live measurement unavailable: {why})"
    if extras and not args.no_energy:
      out['roofline']['energy'] = energy_roofline(local_rank)
    if extras and not args.no_configs:
      # before the 128- and 256-GiB states below: the allocation that follows the release of such a state pays ~6 s of driver
      # work deferred from the release (profiles/r05/alloc_second_buffer_ab.txt) -- rounds 4 and 5 reported it as this entry's
      # "first run"
      out['single_shot_ms'] = single_shot(local_rank)
    if extras and not args.no_ladder_base:
      out['ladder_base'] = ladder_base(local_rank, fusion)
    if extras and not args.no_configs:
      out['configs'] = other_configs(local_rank)
    if not args.no_cpu_baseline and world == 1:
      out['cpu_baseline'] = cpu_baseline(args, ops, g8)
      out['cpu_baseline']['gpu_over_this_baseline'] = out['whole_state_gate_applies_per_s'] / out['cpu_baseline']['value']
    print(json.dumps(out))


if __name__ == '__main__':
  main()
