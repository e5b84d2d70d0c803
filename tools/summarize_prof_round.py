#!/usr/bin/env python
"""gpurun_out/prof_<round> (tools/prof_round.sh) -> the tracked summaries of profiles/<round>/:
  kernel_stats.csv          rocprofv3 --kernel-trace --stats of the default bench command
  kernel_durations.json     per-kernel launch durations from the same trace (first / last launches apart: clock ramp)
  pmc_calibration.json      FETCH_SIZE / WRITE_SIZE of known copy / read / fill kernels -> correction factors
  pmc_summary.json          per bench section: counters per launch of every kernel + HBM bytes per launch, corrected with
                            the calibration factors (keys as bench.py's pmc_traffic() reads them)"""
import collections, csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def short(name):
    n = name.replace("void ", "").replace("sjmi::", "")
    return n.split("(")[0][:80]


def counters(d):
    """-> {kernel: {counter: [values per dispatch]}}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)  # (dispatch, kernel, counter) -> summed over instances
        for r in csv.DictReader(open(f)):
            per[(r.get("Dispatch_Id", "0"), short(r.get("Kernel_Name", "")), r["Counter_Name"])] += float(r["Counter_Value"])
        for (disp, k, c), v in sorted(per.items(), key=lambda kv: int(kv[0][0])):
            agg[k][c].append(v)
    return agg


for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0))
durations = {}
for k, v in dur.items():
    if not (k.startswith("k_") or "k_stage1" in k):
        continue
    d = [x[1] for x in sorted(v)]
    durations[k] = {"launches": len(d), "mean_us": round(sum(d) / len(d), 2), "min_us": round(min(d), 2), "max_us": round(max(d), 2),
                    "first_launches_us": [round(x, 1) for x in d[:60]], "last_launches_us": [round(x, 1) for x in d[-30:]]}
json.dump(durations, open(os.path.join(dst, "kernel_durations.json"), "w"), indent=1)

# ---- calibration: counters are in KiB ----
N = 1 << 30
cal = {"known_bytes": N, "note": "FETCH_SIZE / WRITE_SIZE are reported in KiB; factor = known bytes / (counter * 1024)"}
fetch, write = counters("cal_FETCH_SIZE"), counters("cal_WRITE_SIZE")


def pick(agg, pat, ctr):
    for k, v in agg.items():
        if pat in k.lower() and ctr in v and max(v[ctr]) > 0:
            return k, v[ctr]
    return None, []


for name, pat in (("copy", "copy"), ("read", "reduce"), ("fill", "fill")):
    kf, vf = pick(fetch, pat, "FETCH_SIZE")
    kw, vw = pick(write, pat, "WRITE_SIZE")
    cal[name] = {"kernel_fetch_pass": kf, "FETCH_SIZE_KiB": vf[-3:], "kernel_write_pass": kw, "WRITE_SIZE_KiB": vw[-3:]}
f_copy = [x for x in cal["copy"]["FETCH_SIZE_KiB"] if x > 0]
f_read = [x for x in cal["read"]["FETCH_SIZE_KiB"] if x > 0]
w_copy = [x for x in cal["copy"]["WRITE_SIZE_KiB"] if x > 0]
w_fill = [x for x in cal["fill"]["WRITE_SIZE_KiB"] if x > 0]
cal["fetch_factor_copy"] = round(N / (sum(f_copy) / len(f_copy) * 1024), 4) if f_copy else None
cal["fetch_factor_read"] = round(N / (sum(f_read) / len(f_read) * 1024), 4) if f_read else None
cal["write_factor_copy"] = round(N / (sum(w_copy) / len(w_copy) * 1024), 4) if w_copy else None
cal["write_factor_fill"] = round(N / (sum(w_fill) / len(w_fill) * 1024), 4) if w_fill else None
ff = cal["fetch_factor_copy"] or cal["fetch_factor_read"] or 2.0
wf = cal["write_factor_copy"] or cal["write_factor_fill"] or 1.0
cal["applied"] = {"fetch_factor": ff, "write_factor": wf,
                  "guide_rule": "MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of a wide coalesced read on gfx950 (factor 2), WRITE_SIZE uncalibrated"}
json.dump(cal, open(os.path.join(dst, "pmc_calibration.json"), "w"), indent=1)

# ---- per section ----
KEY = {"main": "stage1_twitter_4g", "x1024": "stage1_twitter_x1024", "unescape": "unescape_twitter_x1024", "synth": "stage1_synthetic_4g",
       "batch": "batch_1m_docs"}
WANT = {"main": ["k_stage1"], "x1024": ["k_stage1"], "synth": ["k_stage1"], "unescape": ["k_strings"],
        # (round 5: the batch step holds no plain k_stage1 launch any more -- the section's k_stage1 dispatches are the primary
        #  workload's parity-check launches and must not be counted)
        "batch": ["k_doc_", "k_strings", "k_stage1_batch", "k_split", "k_batch", "k_tape_", "k_coop", "k_tok_", "k_slow_"]}
summary = {}
for sec, key in KEY.items():
    f, w = counters("pmc_%s_FETCH_SIZE" % sec), counters("pmc_%s_WRITE_SIZE" % sec)
    sq = counters("pmc_%s_SQ" % sec)
    kernels = {}
    tot_f = tot_w = tot_f_raw = 0.0
    for k in sorted(set(f) | set(w)):
        if not any(p in k for p in WANT[sec]):
            continue
        vf, vw = f.get(k, {}).get("FETCH_SIZE", []), w.get(k, {}).get("WRITE_SIZE", [])
        # the LAST dispatches are the timed ones (earlier ones: parity check launch, warmup; for k_stage1 of an extra
        # section the first dispatch is the primary workload's check launch).  k_stage1 in the batch section: two launches
        # per step, the plain pass and the (skipped, empty) parity pass of the sanitized copy -- a whole step = the last two
        per_step = 1  # (round 4: the batch's plain pass is its own kernel, k_stage1_batch; the skipped parity pass k_stage1)
        lf, lw = vf[-2 * per_step:], vw[-2 * per_step:]
        if per_step == 2:
            lf, lw = [sum(lf) / 2.0] if lf else [], [sum(lw) / 2.0] if lw else []
        af, aw = (sum(lf) / len(lf) if lf else 0.0), (sum(lw) / len(lw) if lw else 0.0)
        kernels[k] = {"FETCH_SIZE_KiB_per_launch": round(af, 1), "WRITE_SIZE_KiB_per_launch": round(aw, 1), "dispatches_seen": len(vf),
                      "fetch_bytes_corrected": int(af * 1024 * ff), "write_bytes_corrected": int(aw * 1024 * wf)}
        if k in sq:
            kernels[k]["sq_per_launch"] = {c: round(sum(v[-2:]) / len(v[-2:]), 1) for c, v in sq[k].items()}
        tot_f += af * 1024 * ff
        tot_w += aw * 1024 * wf
        # (kernels whose loads are narrow gathers -- the walkers, the per-boundary pass -- report FETCH_SIZE at factor ~1.0, like the
        #  calibration's narrow read: for them the raw counter is the better figure, the corrected one an upper bound)
        narrow = any(p in k for p in ("k_tok_", "k_coop", "k_doc_prepare", "k_tape_", "k_batch"))
        tot_f_raw += af * 1024 * (1.0 if narrow else ff)
    summary[key] = {"kernels": kernels,
                    "hbm_traffic_bytes_per_launch": {"fetch_bytes": int(tot_f), "write_bytes": int(tot_w), "total": int(tot_f + tot_w),
                                                     "total_lower_bound": int(tot_f_raw + tot_w),
                                                     "fetch_factor": ff, "write_factor": wf,
                                                     "note": "sum over the kernels of one step of this section; counters in KiB x calibration factor; total = every FETCH "
                                                             "counter x the wide-read factor (upper bound), total_lower_bound = raw counter for the kernels with narrow loads"}}
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import csrc_digest  # (the sources these counters were collected from: bench.py's traffic_stale compares)
summary["_collected"] = {"csrc_digest": csrc_digest.csrc_digest()}
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: v["hbm_traffic_bytes_per_launch"]["total"] for k, v in summary.items() if not k.startswith("_")}))
print(json.dumps(cal["applied"]))
