#!/bin/bash
# A/B of a library variant: usage r02_variant_ab.sh /path/to/lib.so   (timings only; a variant may compute garbage)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for lib in "" "$1"; do
    echo "== lib=${lib:-default}"
    QCC_HIP_LIB=$lib bash $R/tools/trace_sweeps.sh 2>&1 | tail -1
  done
done
