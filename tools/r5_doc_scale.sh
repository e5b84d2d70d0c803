#!/bin/bash
# How much of the batch walker's time is per-DOCUMENT cost (and lane under-use at document ends)?  The same bytes as documents 1x / 2x / 4x
# as long (tools/docgen.c -DDOC_SCALE): per-kernel times under rocprofv3.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
for sc in 1 2 4; do
  docs=$((1000000 / sc))
  if [ $sc != 1 ]; then export SJMI_DOCGEN_LIB=$R/tools/variants/libdocgen_x$sc.so SJMI_DOC_SCALE=$sc; fi
  out=gpurun_out/prof_scale_$sc; rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 2 --warmup 1 --preheat 0 --batch-steps 10 --sample 300 --reps 64 --docs $docs > $out/run.log 2>&1
  echo "scale $sc docs $docs rc=$?"
  python - $out <<'PY'
import csv, sys, json
out = sys.argv[1]
for r in sorted(csv.DictReader(open(out + '/t_kernel_stats.csv')), key=lambda r: -float(r['TotalDurationNs'])):
    n = r['Name'].replace('void ', '').replace('sjmi::', '').split('(')[0]
    if n.startswith('k_') and float(r['AverageNs']) > 20000:
        print("   %-36s avg %9.1f us" % (n[:36], float(r['AverageNs']) / 1e3))
for l in open(out + '/run.log'):
    if l.startswith('{'):
        b = json.loads(l)['extra']['batch_1m_docs']; print('   batch ms', b['ms_per_batch'], b['counts']['structurals'], b['counts']['tape_words'])
PY
done
