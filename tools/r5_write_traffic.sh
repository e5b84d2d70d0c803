#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the batch walker (and the batch) for library variants: args = variant names (base = in-tree)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
for v in "$@"; do
  if [ "$v" != "base" ]; then export SJMI_LIB=$R/tools/variants/libsjmi_$v.so; else unset SJMI_LIB; fi
  for ctr in WRITE_SIZE FETCH_SIZE; do
    out=gpurun_out/wt_${v}_$ctr; rm -rf $out; mkdir -p $out
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out -o p -- python bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 2 --warmup 1 --preheat 0 --batch-steps 2 --sample 100 --reps 64 > $out/run.log 2>&1
    python - $out $ctr $v <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != sys.argv[2]: continue
        k = r.get("Kernel_Name", "").split("(")[0].replace("void ", "").replace("sjmi::", "")
        a = agg[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(sys.argv[3], sys.argv[2], {k[:16]: round(v[0] / v[1] * 1024 / 1e6, 1) for k, v in agg.items() if v[0] / max(v[1], 1) > 1000 and k.startswith("k_")}, "MB per launch (raw counter x KiB)")
PY
  done
  unset SJMI_LIB
done
