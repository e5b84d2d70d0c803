#!/bin/bash
# rocprofv3 kernel trace of the device-resident batch parse (tools/walk_bench.py); run on the GPU box via gpurun.
# Usage: tools/prof_batch.sh [n_docs]   -> gpurun_out/prof_batch/kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_batch
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 110 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $R/tools/walk_bench.py ${1:-200000} > $out/trace.log 2>&1
echo "rocprofv3 rc=$?"
tail -2 $out/trace.log
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && cut -d, -f1-8 $out/kernel_stats.csv | cut -c1-170 | head -24
# counter passes (kernel trace only, one counter group per pass), a few launches each
if [ "$2" = "pmc" ]; then
  for grp in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "mem:SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
             "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    WALK_BENCH_ITERS=3 timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d $out/pmc_$name -o p -- python $R/tools/walk_bench.py ${1:-200000} > $out/pmc_$name.log 2>&1
    echo "pmc $name rc=$?"
  done
  python - <<PY
import csv, glob, collections
for d in ("sq", "mem", "fetch", "write"):
    for f in sorted(glob.glob("$out/pmc_%s/**/*counter_collection.csv" % d, recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "").split("(")[0].replace("void ", "").replace("sjmi::", "")
            if k.startswith("k_doc_walk") or k.startswith("k_str_"):
                a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, v in agg.items():
            print(d, k, {c: round(x[0] / max(x[1], 1), 1) for c, x in v.items()})
PY
fi
