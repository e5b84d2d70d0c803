#!/bin/bash
# where a 125k-document step spends its 0.63 ms: kernel trace of bench.py's batch section at --docs 125000 (timeline of one step)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
out=gpurun_out/prof_small
rm -rf $out; mkdir -p $out
B="python bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 2 --warmup 1 --preheat 0 --batch-steps 20 --sample 200 --reps 64 --docs 125000"
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- $B > $out/run.log 2>&1
python - $out <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1] + '/t_kernel_trace.csv'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last full step: from the last k_batch_sep_check to the end
idx = [i for i, r in enumerate(rows) if 'k_batch_sep_check' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('void ', '').replace('sjmi::', '').split('(')[0][:34]
    print("%-36s start %8.1f us  dur %7.1f  gap %6.1f" % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    busy += e - s
    prev_end = e
print("step %.1f us, kernels busy %.1f us" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, busy / 1e3))
PY
