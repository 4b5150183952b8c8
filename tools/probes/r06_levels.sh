#!/bin/bash
# round 6: the level search (planner.h search_levels, a portfolio of host threads) against the tile search it replaced
# (A = the library of commit 2257773 built as _ab/libqcc_hip_r06a.so, if present): supremacy-30 seeds 0..11, sweeps and ms per
# circuit, interleaved fresh processes; planning wall time of the new search on this host (plan of a dry handle).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06lv; mkdir -p $O
cd $R
SEEDS=${SEEDS:-"0 1 2 3 4 5 6 7 8 9 10 11"}
: > $O/ab.txt
for round in 1 2; do
  for seed in $SEEDS; do
    for v in old new; do
      unset QCC_HIP_LIB
      if [ $v = old ]; then [ -f $R/_ab/libqcc_hip_r06a.so ] || continue; export QCC_HIP_LIB=$R/_ab/libqcc_hip_r06a.so; fi
      echo "## $v seed $seed round $round" >> $O/ab.txt
      timeout 120 python tools/run_workload.py sup30s$seed 14 2>&1 | grep -E "step ms|'sweeps'" | tail -2 >> $O/ab.txt
    done
  done
done
unset QCC_HIP_LIB
python3 - <<'PY' | tee gpurun_out/r06lv/summary.txt
import re, collections, statistics
rows = collections.defaultdict(list)
nsw = {}
key = None
for ln in open('gpurun_out/r06lv/ab.txt'):
    if ln.startswith('## '):
        p = ln.split(); key = (int(p[3]), p[1])
    elif "'sweeps'" in ln:
        nsw[key] = int(re.search(r"'sweeps': (\d+)", ln).group(1)) // 15
    elif 'step ms' in ln:
        v = [float(x) for x in ln.split('step ms')[1].split()]
        rows[key].append((statistics.median(v[-5:]), max(v[-5:])))
for k in sorted(rows):
    print('seed', k[0], f'{k[1]:4s}', nsw.get(k), 'sweeps', ' '.join(f'{x:.2f} (max {m:.2f})' for x, m in rows[k]), 'ms per circuit (median and maximum of the last 5 of 14 steps -- plan cache on: GPU time --, one figure per process)')
PY
python3 - <<'PY' | tee -a gpurun_out/r06lv/summary.txt
import ctypes, json, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qcc_amd import native, workloads
lib = native.load()
print('planning wall time of one flush on this host (%d cpus), dry handle, ms: sweeps' % os.cpu_count())
for seed in range(12):
    ops, g8 = workloads.supremacy_stream(30, 20, seed=seed).arrays()
    g8 = np.ascontiguousarray(g8, dtype=np.float64)
    ts = []
    for rep in range(3):
        h = ctypes.c_void_p()
        native.check(lib.qh_create_dry(30, 128, ctypes.byref(h)))
        native.check(lib.qh_set_fusion(h, native.QH_FUSE_SWEEP))
        dp = ctypes.POINTER(ctypes.c_double)
        for k in range(len(ops)):
            gp = ctypes.cast(g8.ctypes.data + 64 * k, dp)
            c, t = int(ops[k, 0]), int(ops[k, 1])
            native.check(lib.qh_apply1(h, t, gp) if c == workloads.NO_CTL else lib.qh_applyc(h, c, t, gp))
        need = ctypes.c_uint64()
        t0 = time.perf_counter()
        lib.qh_plan_json(h, None, 0, ctypes.byref(need))
        ts.append((time.perf_counter() - t0) * 1e3)
        buf = ctypes.create_string_buffer(need.value)
        lib.qh_plan_json(h, buf, need.value, None)
        lib.qh_destroy(h)
    print('  seed %2d: %s ms: %d sweeps' % (seed, ' '.join('%.1f' % t for t in ts), len(json.loads(buf.value.decode())['sweeps'])))
PY
