#!/bin/bash
# SQ counters of k_stage1 on twitter x1024 for library variants ("cur" = in-tree): instructions per launch -> per 4 KiB wave-step
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = cur ]; then unset SJMI_LIB; else export SJMI_LIB=$R/tools/variants/libsjmi_$v.so; fi
  out=$R/gpurun_out/pmc_s1_$v; rm -rf $out
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out -o p -- python $R/bench.py --reps 1024 --no-cpu-baseline --no-extras --steps 3 --warmup 1 --preheat 0 > $out.log 2>&1
  python - "$out" "$v" <<'PY'
import sys, glob, csv, collections
d, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_stage1" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k in acc:
    steps = 659732480 / 4096  # wave-steps of twitter x1024
    c = {a: b / n[k] for a, b in acc[k].items()}
    print(v, k[:40], "launches", n[k], "VALU/step %.1f" % (c["SQ_INSTS_VALU"] / steps), "SALU/step %.1f" % (c["SQ_INSTS_SALU"] / steps),
          "wave_cycles/step %.0f" % (c["SQ_WAVE_CYCLES"] * 4 / steps), "wait_any %.2f" % (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]),
          "wait_inst %.2f" % (c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]), "active_valu %.2f" % (c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]))
PY
done
