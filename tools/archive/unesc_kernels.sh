# per-kernel times of the unescape and batch bench sections (rocprofv3 kernel trace); run on the GPU box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/unk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/unk -o t -- python bench.py --no-cpu-baseline --sections unescape,batch --skip-main-timing --steps 3 --warmup 1 --preheat 0 --batch-steps 5 > gpurun_out/unk.log 2>&1
python - <<'PY'
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/unk/t_kernel_trace.csv')):
    n = r['Kernel_Name'].split('(')[0].replace('void sjmi::', '').replace('sjmi::', '')
    if n.startswith('k_'):
        d[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in sorted(d.items()):
    v.sort()
    print("%-28s n %4d  min %8.1f  median %8.1f  max %8.1f us" % (n, len(v), v[0], v[len(v) // 2], v[-1]))
PY
