#!/bin/bash
# per-kernel times of a SMALL batch (one rank's share of a strong-scaled run): $1 = documents
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
out=gpurun_out/prof_small_$1
rm -rf $out; mkdir -p $out
B="python bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 2 --warmup 1 --preheat 0 --batch-steps 40 --sample 500 --reps 64 --docs $1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- $B > $out/run.log 2>&1
echo "rc=$?"
python - $out <<'PY'
import csv, sys, json
out = sys.argv[1]
rows = [r for r in csv.DictReader(open(out + '/t_kernel_stats.csv'))]
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    n = r['Name'].replace('void ', '').replace('sjmi::', '').split('(')[0]
    if n.startswith('at::') or 'elementwise' in n or 'Cijk' in n: continue
    print("%-44s calls %5s avg %9.1f us  min %9.1f  max %9.1f" % (n[:44], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
# gaps between consecutive kernels of one step (from the trace): start(k+1) - end(k)
tr = [r for r in csv.DictReader(open(out + '/t_kernel_trace.csv'))]
tr.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('void ', '').replace('sjmi::', '').split('(')[0] for r in tr]
# last full step: find the last k_stage1_batch
idx = [i for i, n in enumerate(names) if n.startswith('k_stage1_batch')]
if len(idx) >= 3:
    a, b = idx[-3], idx[-2]
    t0 = int(tr[a]['Start_Timestamp'])
    for i in range(a, b + 1):
        print("  %-36s start %8.1f us  dur %8.1f us" % (names[i][:36], (int(tr[i]['Start_Timestamp']) - t0) / 1e3, (int(tr[i]['End_Timestamp']) - int(tr[i]['Start_Timestamp'])) / 1e3))
for l in open(out + '/run.log'):
    if l.startswith('{'):
        d = json.loads(l)
        b = d['extra']['batch_1m_docs']
        print('batch', b['value'], b['ms_per_batch'])
PY
