#!/bin/bash
# rocprofv3 of the streaming string pass on twitter x1024 (tools/unescape_prof.py): kernel trace + stats, then the SQ and
# the HBM counters in passes of their own.  Run on the GPU box: gpurun -- tools/strings_prof.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-strings}
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P="python $R/tools/unescape_prof.py 1024 5"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $P > $out/trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/sq -o p -- $P > $out/sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $out/sq2 -o p -- $P > $out/sq2.log 2>&1; echo "sq2 rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/$ctr -o p -- $P > $out/$ctr.log 2>&1; echo "$ctr rc=$?"
done
python - <<PY
import csv, glob, collections
for f in glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sjmi" in r["Name"] or "k_" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
for d in ("sq", "sq2", "FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])] += float(r["Counter_Value"])
        for (disp, k, c), v in sorted(per.items()):
            agg[k][c].append(v)
    for k, cs in agg.items():
        if "k_strings" in k or "k_stage1" in k or "k_str_" in k:
            print(d, k, {c: round(sum(v[-3:]) / len(v[-3:]), 1) for c, v in cs.items()})
PY
