#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
out=gpurun_out/prof_tl
rm -rf $out; mkdir -p $out
python tools/single_doc_timeline.py 2>/dev/null | tail -1
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- python tools/single_doc_timeline.py > $out/run.log 2>&1
python - $out <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1] + '/t_kernel_trace.csv')):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('sjmi::', '').split('(')[0][:36]))
for r in csv.DictReader(open(sys.argv[1] + '/t_memory_copy_trace.csv')):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Name', ''))[:24]))
ev.sort()
# one call in the middle of the run: find the H2D copies and take the span between two consecutive ones near the end
h2d = [i for i, e in enumerate(ev) if 'HOST_TO_DEVICE' in e[2].upper() or 'HtoD' in e[2]]
a, b = h2d[-3], h2d[-2]
t0 = ev[a][0]
prev = t0
for s, e, n in ev[a:b]:
    print("%-42s start %7.1f us dur %6.1f gap %6.1f" % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = e
print("call period %.1f us" % ((ev[b][0] - t0) / 1e3))
PY
