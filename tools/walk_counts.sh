#!/bin/bash
# Dynamic instruction counts of the batch pipeline's kernels (rocprofv3 --pmc, kernel trace only) on bench.py's
# 1,000,000-document batch, then the walker under its ablation switches.  Run on the GPU box via gpurun.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
out=gpurun_out/walk_counts
rm -rf $out; mkdir -p $out
B="python bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 2 --warmup 1 --preheat 0 --batch-steps 2"
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d $out/sq -o p -- $B > $out/sq.log 2>&1
echo "pmc rc=$?"
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/walk_counts/sq/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0].replace("void ", "").replace("sjmi::", "")
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in sorted(agg.items()):
        m = {c: x[0] / max(x[1], 1) for c, x in v.items()}
        if m.get("SQ_INSTS_VALU", 0) > 1e6:
            print(k[:28], {c: "%.3g" % x for c, x in m.items()})
PY
if [ "$1" = "abl" ]; then
for a in 0 1 2 4 7; do
  rm -rf gpurun_out/abl_$a
  SJMI_COOP_ABLATE=$a timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$a -o t -- $B > gpurun_out/abl_$a.log 2>&1
  python - $a <<'PY'
import csv, sys
a = sys.argv[1]
for r in csv.DictReader(open('gpurun_out/abl_%s/t_kernel_stats.csv' % a)):
    if 'k_coop_walk' in r['Name'] and int(r['Calls']) < 20:
        print("ablate", a, r['Name'][:40], r['Calls'], "avg %.1f us" % (float(r['AverageNs']) / 1e3))
PY
done
fi
