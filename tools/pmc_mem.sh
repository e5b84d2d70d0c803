#!/bin/bash
# memory-pipeline counters of the batch walker (k_tok_*) (VMEM issue, LDS), tools/batch_nocheck.py; run on the GPU box: tools/pmc_mem.sh lib ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
K=${PMC_KERNEL:-k_tok}
for lib in "$@"; do
  if [ "$lib" = base ]; then unset SJMI_LIB; else export SJMI_LIB=$R/tools/variants/libsjmi_$lib.so; fi
  # (TA_* and TCP_* counter sets hung rocprofv3 on these boxes -- two 200 s timeouts -- and were taken out)
  for set in \
             "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pm_$lib
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pm_$lib -o p -- python $R/tools/batch_nocheck.py 1000000 2 > /tmp/pm_$lib.log 2>&1
  rc=$?
  python - /tmp/pm_$lib $K $lib $rc <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0].replace("void ", "").replace("sjmi::", "")
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in sorted(agg.items()):
        m = {c: x[0] / max(x[1], 1) for c, x in v.items()}
        if k.startswith(sys.argv[2]):
            print(sys.argv[3], k[:24], {c.replace("_sum", ""): "%.4g" % x for c, x in m.items()})
PY
  [ $rc != 0 ] && { echo "rc=$rc"; tail -3 /tmp/pm_$lib.log | cut -c1-300; }
  done
done
