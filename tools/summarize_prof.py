#!/usr/bin/env python
"""Turn a tools/archive/prof.sh output directory (gpurun_out/prof_<tag>) into the tracked summaries under profiles/<round>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats), kernel_durations.json (per-launch durations of k_stage1 from the
kernel trace, so the steady state can be told from the clock ramp), pmc_summary.json (per-launch counter averages)."""
import collections, csv, glob, json, os, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in csv.DictReader(open(f)) if "k_stage1" in r["Kernel_Name"]]
    s = sorted(d[-100:])
    json.dump({"kernel": "k_stage1", "launches": len(d), "mean_us_all": round(sum(d) / len(d), 2),
               "last_100_launches_us": {"mean": round(sum(s) / len(s), 2), "p10": s[10], "p50": s[50], "p90": s[90], "min": s[0], "max": s[-1]},
               "first_150_launches_us": [round(x, 1) for x in d[:150]]}, open(os.path.join(dst, "kernel_durations.json"), "w"), indent=1)
out = {}
for grp in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_mem"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(src, grp, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_stage1" in r.get("Kernel_Name", ""):
                a = agg[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    out[grp] = {k: {"avg_per_launch": v[0] / max(v[1], 1), "launches": v[1]} for k, v in sorted(agg.items())}
fetch = out["pmc_fetch"]["FETCH_SIZE"]["avg_per_launch"] * 1024 * 2
write = out["pmc_write"]["WRITE_SIZE"]["avg_per_launch"] * 1024
out["hbm_traffic_bytes_per_launch"] = {
    "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "total": fetch + write,
    "note": "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM) so it is doubled"}
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out["hbm_traffic_bytes_per_launch"]))
