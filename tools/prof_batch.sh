#!/bin/bash
# per-kernel times of the configs[3] batch (bench.py's batch section) under rocprofv3 --kernel-trace --stats; run on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
out=gpurun_out/prof_batch_${1:-x}
rm -rf $out; mkdir -p $out
B="python bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 2 --warmup 1 --preheat 0 --batch-steps 10 --sample 2000 --reps 64"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- $B > $out/run.log 2>&1
echo "rc=$?"
python - $out <<'PY'
import csv, sys, json
out = sys.argv[1]
rows = [r for r in csv.DictReader(open(out + '/t_kernel_stats.csv'))]
tot = 0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    n = r['Name'].replace('void ', '').replace('sjmi::', '').split('(')[0]
    if n.startswith('at::') or 'elementwise' in n or 'Cijk' in n: continue
    print("%-44s calls %5s avg %9.1f us  min %9.1f  max %9.1f" % (n[:44], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
for l in open(out + '/run.log'):
    if l.startswith('{'):
        d = json.loads(l)
        b = d['extra']['batch_1m_docs']
        print('batch', b['value'], b['ms_per_batch'], b['counts'])
PY
