#!/bin/bash
# k_tok_walk with parts of its work compiled out (-DSJMI_TOK_ABL=bits, tools/build_variant.sh): per-kernel times under rocprofv3
# usage (GPU box): tools/abl_tok.sh lib1 lib2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export SJMI_LIB=$R/tools/variants/libsjmi_$lib.so
  rm -rf /tmp/abl_$lib
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$lib -o t -- python $R/tools/batch_nocheck.py 1000000 8 > /tmp/abl_$lib.log 2>&1
  echo "== $lib rc=$? $(grep 'ms per step' /tmp/abl_$lib.log)"
  python - /tmp/abl_$lib <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('void ', '').replace('sjmi::', '').split('(')[0]
    if n.startswith('k_tok') or n.startswith('k_coop') or n.startswith('k_strings') or n.startswith('k_stage1_b'):
        print("   %-34s calls %4s avg %9.1f us min %9.1f" % (n[:34], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
