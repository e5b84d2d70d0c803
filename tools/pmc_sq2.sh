#!/bin/bash
# more SQ / SQC counters of the batch walker (k_tok_*) (instruction fetch, scalar cache, levels); run on the GPU box: tools/pmc_sq2.sh lib ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = base ]; then unset SJMI_LIB; else export SJMI_LIB=$R/tools/variants/libsjmi_$lib.so; fi
  for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_THREAD_CYCLES_VALU"; do
  rm -rf /tmp/pq_$lib
  timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pq_$lib -o p -- python $R/tools/batch_nocheck.py 1000000 2 > /tmp/pq_$lib.log 2>&1
  rc=$?
  python - /tmp/pq_$lib k_tok $lib <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0].replace("void ", "").replace("sjmi::", "")
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in sorted(agg.items()):
        m = {c: x[0] / max(x[1], 1) for c, x in v.items()}
        if k.startswith(sys.argv[2]):
            print(sys.argv[3], {c: "%.4g" % x for c, x in m.items()})
PY
  [ $rc != 0 ] && { echo "rc=$rc"; tail -2 /tmp/pq_$lib.log | cut -c1-200; }
  done
done
