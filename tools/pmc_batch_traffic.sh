#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the batch walker for library variants (raw counters, KiB x 1024; own passes): tools/pmc_batch_traffic.sh name[:lib] ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec#*:}; [ "$lib" = "$spec" ] && lib=""
  [ -n "$lib" ] && export SJMI_LIB=$R/tools/variants/libsjmi_$lib.so || unset SJMI_LIB
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pt_$name
    timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d /tmp/pt_$name -o p -- python $R/tools/batch_nocheck.py 1000000 2 > /tmp/pt_$name.log 2>&1
    python - /tmp/pt_$name $name $ctr <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r.get("Dispatch_Id", "0"), r.get("Kernel_Name", "").split("(")[0].replace("void ", "").replace("sjmi::", ""))] += float(r["Counter_Value"])
    last = {}
    for (d, k), v in sorted(per.items(), key=lambda kv: int(kv[0][0])):
        last[k] = v
    for k, v in sorted(last.items()):
        if k.startswith(("k_tok", "k_strings", "k_stage1_batch", "k_doc_prepare")):
            print("%-8s %-12s %-32s %9.1f MB" % (sys.argv[2], sys.argv[3], k[:32], v * 1024 / 1e6))
PY
  done
done
