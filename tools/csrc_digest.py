"""A digest of the product's kernel sources (simdjson-java_amd/csrc/**, include/sjmi.h): stamped into profiles/<round>/pmc_summary.json
when the counters are collected (tools/summarize_prof_round.py, on the GPU box) and compared by bench.py with the sources it runs
from -- `roofline.traffic` is a constant read from that file, and `traffic_stale` says when the kernels have changed since.
(No git needed: the GPU box has no .git.)"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_files(root=ROOT):
    base = os.path.join(root, "simdjson-java_amd", "csrc")
    out = []
    for d, _, fs in os.walk(base):
        for f in fs:
            if f.endswith((".hip", ".h", ".cpp")):
                out.append(os.path.join(d, f))
    out.append(os.path.join(root, "include", "sjmi.h"))
    return sorted(out)


def csrc_digest(root=ROOT):
    h = hashlib.sha256()
    for p in csrc_files(root):
        h.update(os.path.relpath(p, root).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()[:16]


def traffic_stale(summary, root=ROOT):
    """-> (stale, why).  summary = the parsed pmc_summary.json.  Stale when the digest stamped at collection differs from the
    sources' -- or when there is no stamp (summaries older than round 6) and git says csrc/ changed after the stamped commit, or
    cannot say."""
    meta = summary.get("_collected", {}) if isinstance(summary, dict) else {}
    want = meta.get("csrc_digest")
    if want:
        have = csrc_digest(root)
        return (have != want, "csrc digest %s, summary collected at %s" % (have, want))
    commit = meta.get("commit")
    if commit and os.path.isdir(os.path.join(root, ".git")):
        import subprocess
        try:
            last = subprocess.check_output(["git", "-C", root, "log", "-1", "--format=%H", "--", "simdjson-java_amd/csrc", "include/sjmi.h"],
                                           stderr=subprocess.DEVNULL).decode().strip()
            dirty = subprocess.call(["git", "-C", root, "diff", "--quiet", "--", "simdjson-java_amd/csrc", "include/sjmi.h"]) != 0
            anc = subprocess.call(["git", "-C", root, "merge-base", "--is-ancestor", last, commit], stderr=subprocess.DEVNULL) == 0
            return (dirty or not anc, "last csrc commit %s %s the summary's commit %s%s" % (last[:7], "is at or before" if anc else "is NOT an ancestor of",
                                                                                       commit, ", working tree dirty" if dirty else ""))
        except (OSError, subprocess.CalledProcessError):
            pass
    return (True, "the summary carries no csrc digest and git cannot tell: treated as stale")


if __name__ == "__main__":
    print(csrc_digest())
