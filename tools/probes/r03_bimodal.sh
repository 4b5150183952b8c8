cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
# per-process variation of the QFT-30 step time: 16 fresh processes, 8 steps each; buffer addresses printed
for i in $(seq 1 16); do
  echo "## proc $i" >> $O/procs.txt
  QH_SWEEP_TIMING=1 timeout 200 python tools/run_workload.py qft30 8 2>&1 | grep -a "qh sweeps" | tail -6 >> $O/procs.txt
done
python3 - <<'PY'
import re, statistics
cur=None; rows={}
for l in open('gpurun_out/r03q/procs.txt'):
    if l.startswith('##'): cur=l.split()[2]; rows[cur]=[]
    else:
        m=re.search(r'psi=(\S+) alt=(\S+)\]',l); v=[float(x) for x in re.findall(r'[0-9.]+',l.split(']')[1])]
        rows[cur].append((m.group(1),m.group(2),sum(v),v))
for k,r in rows.items():
    print(k, r[-1][0], r[-1][1], 'step ms', [round(x[2],2) for x in r], 'median', round(statistics.median(x[2] for x in r),3))
PY
