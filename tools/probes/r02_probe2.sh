#!/bin/bash
# Round-2 probe 2 (GPU box): workgroup shape / rotation for out-of-place tile geometries, and the
# address-translation counters of the in-place geometries.  Writes gpurun_out/r02b/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C=3,4,5,6,7,8,9,10,11
M=12,13,14,15,16,17,18,19,20
H=21,22,23,24,25,26,27,28,29
ARGS=""
for f in u1 u2 u4 u2r3 u4r3; do
  ARGS="$ARGS $C:$C:i$f $M:$M:i$f $H:$H:i$f $C:$C:$f $C:$M:$f $C:$H:$f $M:$C:$f $H:$C:$f"
done
timeout 600 $R/tools/membench/oopsweep 30 $ARGS > $O/oopsweep_shapes.txt 2>&1
cat $O/oopsweep_shapes.txt
# translation counters, one small set per pass, on the in-place and out-of-place geometries (u2)
G="$C:$C:iu2 $M:$M:iu2 $H:$H:iu2 $C:$H:u2 $H:$C:u2"
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '+' | cut -c1-60)
  rm -rf /tmp/pm
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -o t -- $R/tools/membench/oopsweep 30 $G > /tmp/pm.log 2>&1
  f=$(find /tmp/pm -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" "$tag" >> $O/tlb_counters.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
print('##', sys.argv[2])
for i, (d, c) in enumerate(by.items()):
    if i % 6 == 5:   # last repetition of each geometry (6 launches each)
        print('geom', i // 6, {k: int(v) for k, v in c.items()})
PY
  else
    echo "## $tag: no counter file" >> $O/tlb_counters.txt; tail -5 /tmp/pm.log >> $O/tlb_counters.txt
  fi
done
cat $O/tlb_counters.txt
