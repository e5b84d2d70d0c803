#!/bin/bash
# SQ instruction counters of the batch pipeline's kernels (rocprofv3 --pmc, kernel trace only); run on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
out=gpurun_out/pmc_batch_${1:-x}
rm -rf $out; mkdir -p $out
B="python bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 2 --warmup 1 --preheat 0 --batch-steps 2 --sample 100 --reps 64"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d $out -o p -- $B > $out/run.log 2>&1
echo "pmc rc=$?"
python - $out <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0].replace("void ", "").replace("sjmi::", "")
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in sorted(agg.items()):
        m = {c: x[0] / max(x[1], 1) for c, x in v.items()}
        if m.get("SQ_INSTS_VALU", 0) > 1e6 and not k.startswith("at::"):
            print(k[:30], {c.replace("SQ_", "").replace("INSTS_", ""): "%.3g" % x for c, x in m.items()})
PY
