#!/bin/bash
# per-kernel times of the call for rejected batches (one document of 1 M fails stage 1) for library variants: tools/ab_rejected.sh name[:lib] ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec#*:}; [ "$lib" = "$spec" ] && lib=""
  [ -n "$lib" ] && export SJMI_LIB=$R/tools/variants/libsjmi_$lib.so || unset SJMI_LIB
  out=gpurun_out/abr_$name; rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python tools/batch_exact_nocheck.py 1000000 10 bad rejected > $out/run.log 2>&1
  echo "== $name $(grep -h 'ms per step' $out/run.log | cut -c1-24)"
  python - $out <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1] + '/t_kernel_stats.csv'))]
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:6]:
    n = r['Name'].replace('void ', '').replace('sjmi::', '').split('(')[0]
    if 'rocclr' in n: continue
    print("   %-36s avg %8.1f us" % (n[:36], float(r['AverageNs']) / 1e3))
PY
done
