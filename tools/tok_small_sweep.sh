#!/bin/bash
# k_tok_stream at a small batch (one rank's share of a strong-scaled run) for several run sizes: tools/tok_small_sweep.sh <documents> run ...
R=${GRAFT_REPO_ROOT:-/root/repo}
n=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $R
for run in "$@"; do
  rm -rf /tmp/tss; 
  SJMI_TS_RUN_DOCS=$run timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tss -o t -- python tools/batch_nocheck.py $n 40 > /tmp/tss.log 2>&1
  python - $run <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open('/tmp/tss/t_kernel_stats.csv'))]
for r in rows:
    if 'k_tok_stream' in r['Name']: print("run %s: k_tok_stream avg %.1f us min %.1f" % (sys.argv[1], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
