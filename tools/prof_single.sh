# kernel trace of tools/single_doc_modes.py (run on the GPU box): per-kernel durations, twitter.json and the 64 MiB array apart
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_single -o single -- python tools/single_doc_modes.py > gpurun_out/prof_single.log 2>&1
grep "parse" gpurun_out/prof_single.log
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/prof_single/single_kernel_trace.csv')))
small, big = collections.defaultdict(list), collections.defaultdict(list)
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    n = r['Kernel_Name'].split('(')[0][-40:]
    small[n].append(d)
for n, v in sorted(small.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-42s calls %5d  median %9.1f us  max %9.1f us  total %9.1f ms" % (n, len(v), v[len(v)//2] / 1e3, v[-1] / 1e3, sum(v) / 1e6))
PY
