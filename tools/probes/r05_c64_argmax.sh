cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05c64
python -m pytest tests/test_gpu_parity.py -x -q -k "argmax or readers" 2>&1 | grep -a -E "passed|failed|Error|assert" | tail -5
(FUZZ_BW=64 timeout 400 python tools/fuzz_parity.py 300 13001 > gpurun_out/r05c64/f1.txt 2>&1) &
(FUZZ_BW=64 QH_SEATS=2 QH_WAVE_BITS=2 timeout 400 python tools/fuzz_parity.py 300 13002 > gpurun_out/r05c64/f2.txt 2>&1) &
(FUZZ_BW=64 QH_RELAYOUT=0 timeout 400 python tools/fuzz_parity.py 300 13003 > gpurun_out/r05c64/f3.txt 2>&1) &
wait
tail -n 2 gpurun_out/r05c64/f*.txt
QH_SWEEP_TIMING=1 python tools/run_workload.py qft30c64 5 2>&1 | grep -a "qh sweeps" | tail -3
python - <<'PY'
import sys, time
sys.path.insert(0,'.')
import numpy as np
from qcc_amd import device, native, workloads
ops, g8 = workloads.qft_stream(range(30)).arrays()
for bw in (64, 128):
  with device.DeviceState(30, bw, fusion=native.QH_FUSE_SWEEP) as st:
    st.init_basis(5)
    for rep in range(4):
      st.run_stream(ops, g8); st.flush(); st.sync()
      t0=time.perf_counter(); st.argmax(); t1=time.perf_counter()       # queue empty: the full pass
      st.run_stream(ops, g8)
      t2=time.perf_counter(); r=st.argmax(); t3=time.perf_counter()     # behind a flush: sweeps + per-unit maxima
      st.run_stream(ops, g8); st.flush()
      t4=time.perf_counter(); st.sync(); t5=time.perf_counter()
    print('bw', bw, 'argmax alone %.2f ms; qft+argmax %.2f ms; qft alone (flush+sync) %.2f ms' % ((t1-t0)*1e3, (t3-t2)*1e3, (t5-t4)*1e3), r)
PY
