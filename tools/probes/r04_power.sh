#!/bin/bash
# round 4: socket power and shader clock while supremacy-30 loops -- the real islands, the islands without their memory
# instructions (QH_ISLAND_NOMEM builds: timing / power only) -- and while the 30-qubit QFT loops
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04pw; mkdir -p $O
cd $R
for v in real nomem1; do
  unset QCC_HIP_LIB; [ $v != real ] && export QCC_HIP_LIB=$R/tools/probes/variants/libqcc_hip_$v.so
  timeout 120 python tools/probes/smi_trace.py sup30 20 $O/smi_sup30_$v.csv > $O/smi_sup30_$v.log 2>&1
done
unset QCC_HIP_LIB
timeout 120 python tools/probes/smi_trace.py qft30 20 $O/smi_qft30_real.csv > $O/smi_qft30_real.log 2>&1
python3 - <<'PY'
import csv, glob, statistics
for f in sorted(glob.glob('gpurun_out/r04pw/*.csv')):
    rows = list(csv.DictReader(open(f)))
    def col(name):
        out = []
        for r in rows[len(rows)//4:]:
            try: out.append(float(str(r.get(name, '')).split()[0]))
            except Exception: pass
        return out
    pw = col('power1_average') or col('power1_input')
    print(f, 'rows', len(rows), 'power W median', round(statistics.median(pw)/1e6, 1) if pw else None, 'columns', list(rows[0].keys())[:12] if rows else None)
    for name in rows[0].keys() if rows else []:
        if 'sclk' in name.lower():
            v = col(name)
            if v: print('   ', name, 'median', statistics.median(v))
PY
