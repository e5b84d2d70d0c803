#!/usr/bin/env python
"""gpurun_out/profiles_<round> (written on the GPU box by tools/prof_round.sh <round>) -> profiles/<round>/ (tracked), stamped with
the commit the kernels were built from -- bench.py quotes roofline.traffic from here and names file + commit as its source.
usage: install_prof.py r5 [commit]"""
import json, os, shutil, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rd = sys.argv[1]
src, dst = os.path.join(R, "gpurun_out", "profiles_" + rd), os.path.join(R, "profiles", rd)
os.makedirs(dst, exist_ok=True)
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "-C", R, "rev-parse", "--short", "HEAD"]).decode().strip()
for f in ("kernel_stats.csv", "kernel_durations.json", "pmc_calibration.json", "pmc_summary.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
p = os.path.join(dst, "pmc_summary.json")
d = json.load(open(p))
stamp = d.get("_collected", {})
d["_collected"] = {"csrc_digest": stamp.get("csrc_digest"), "commit": commit, "date": time.strftime("%Y-%m-%d"), "by": "tools/prof_round.sh %s on one MI355X via gpurun" % rd,
                   "note": "kernels of this commit; later commits of the round that do not touch csrc/ leave the figures valid"}
json.dump(d, open(p, "w"), indent=1)
print("installed", dst, "commit", commit)
