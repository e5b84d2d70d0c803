#!/bin/bash
# SQ counters of the batch walker (k_tok_stream / k_tok_walk), per launch; run on the GPU box: tools/pmc_tok.sh name[:lib][@ENV=..] ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  envs=""; case $spec in *@*) envs=${spec#*@}; spec=${spec%%@*};; esac
  name=${spec%%:*}; lib=${spec#*:}; [ "$lib" = "$spec" ] && lib=""
  [ -n "$lib" ] && export SJMI_LIB=$R/tools/variants/libsjmi_$lib.so || unset SJMI_LIB
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/pa_$name
  env $envs timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pa_$name -o p -- python $R/tools/batch_nocheck.py 1000000 2 > /tmp/pa_$name.log 2>&1
  python - /tmp/pa_$name $name <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0].replace("void ", "").replace("sjmi::", "")
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in sorted(agg.items()):
        m = {c: x[0] / max(x[1], 1) for c, x in v.items()}
        if k.startswith("k_tok"):
            print(sys.argv[2], k[:14], {c.replace("SQ_", "").replace("INSTS_", ""): "%.4g" % x for c, x in m.items()})
PY
  done
done
