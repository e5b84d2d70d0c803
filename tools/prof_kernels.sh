#!/bin/bash
# per-kernel times of any command under rocprofv3 --kernel-trace --stats (run on the GPU box): tools/prof_kernels.sh <tag> <command ...>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $R
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- "$@" > $out/run.log 2>&1
echo "rc=$?"; tail -3 $out/run.log
python - $out <<'PY'
import csv, sys
out = sys.argv[1]
rows = [r for r in csv.DictReader(open(out + '/t_kernel_stats.csv'))]
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    n = r['Name'].replace('void ', '').replace('sjmi::', '').split('(')[0]
    if n.startswith('at::') or 'elementwise' in n or 'Cijk' in n: continue
    print("%-52s calls %5s avg %9.1f us  min %9.1f  max %9.1f" % (n[:52], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
