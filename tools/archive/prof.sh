#!/bin/bash
# rocprofv3 collection for bench.py (run on the GPU box via gpurun). Usage: tools/prof.sh <tag> [bench args...]
# Writes CSV summaries under gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $R/bench.py --no-cpu-baseline "$@" > $out/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/pmc_sq -o p -- python $R/bench.py --preheat 0 --warmup 1 --steps 3 --no-cpu-baseline "$@" > $out/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $out/pmc_fetch -o p -- python $R/bench.py --preheat 0 --warmup 1 --steps 3 --no-cpu-baseline "$@" > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $out/pmc_write -o p -- python $R/bench.py --preheat 0 --warmup 1 --steps 3 --no-cpu-baseline "$@" > $out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD -d $out/pmc_mem -o p -- python $R/bench.py --preheat 0 --warmup 1 --steps 3 --no-cpu-baseline "$@" > $out/pmc_mem.log 2>&1
find $out -name "*.csv" | head -30
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_stage1" in r["Name"] or "memset" in r["Name"].lower() or "fill" in r["Name"].lower():
            print("stats:", r["Name"][:60], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
for d in ("pmc_sq","pmc_fetch","pmc_write","pmc_mem"):
    for f in sorted(glob.glob("$out/%s/**/*counter_collection.csv" % d, recursive=True)):
        agg = collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if "k_stage1" in r.get("Kernel_Name",""):
                a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        print(d, "(per big launch)", {k: round(v[0]/max(v[1],1),1) for k,v in agg.items()}, "launches", max([v[1] for v in agg.values()] or [0]))
PY
