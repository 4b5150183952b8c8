# A/B of QH_ALLOC_CONTIG on the whole default bench line (every single-GPU config), interleaved, 3 rounds.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q; mkdir -p $O
: > $O/contig_ab.txt
for r in 1 2 3; do
  for c in 1 0; do
    QH_ALLOC_DEBUG=1 QH_ALLOC_CONTIG=$c timeout 900 python bench.py --no-cpu-baseline > $O/b.json 2> $O/b.err
    echo "## round $r QH_ALLOC_CONTIG=$c" >> $O/contig_ab.txt
    grep "qh alloc" $O/b.err | awk '{print $3, $5, $9}' | sort | uniq -c | sort -k2n | head -30 >> $O/contig_ab.txt
    python - $O/b.json >> $O/contig_ab.txt <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('qft30 ms/step', d['ms_per_step'], 'median', d.get('median_ms_per_step'), 'frac', d['roofline']['frac'])
for k, v in d.get('configs', {}).items():
  print(' ', k, {kk: v[kk] for kk in v if kk in ('ms_per_step', 'median_ms_per_step', 'roofline_frac', 'ms_per_iteration')})
lb = d.get('ladder_base')
if lb: print('  ladder_base', {kk: lb[kk] for kk in lb if kk in ('ms_per_step', 'median_ms_per_step', 'roofline_frac')})
print('  single_shot', d.get('single_shot_ms'))
PY
  done
done
cat $O/contig_ab.txt
