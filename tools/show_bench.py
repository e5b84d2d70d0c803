#!/usr/bin/env python
"""prints the figures of a bench.py JSON line (tools/show_bench.py FILE)"""
import json, sys
for l in open(sys.argv[1]):
    l = l.strip()
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    r = d.get("roofline", {})
    print({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "n_gpus") if k in d}, "frac", r.get("frac"),
          "cold", (r.get("cold") or {}).get("frac"), "settled", (r.get("settled") or {}).get("frac"))
    cb = d.get("cpu_baseline", {})
    print("cpu_baseline", cb.get("value"), cb.get("unit"), "one core", cb.get("one_core"), "cores", cb.get("cores"))
    for k, v in d.get("extra", {}).items():
        cb = v.get("cpu_baseline", {})
        rr = v.get("roofline") or {}
        print("%-32s %s %s | frac %s ms %s | cpu %s %s (one core %s)" % (k, v.get("value"), v.get("unit"), rr.get("frac"),
              rr.get("avg_kernel_ms", rr.get("avg_ms_per_call", v.get("ms_per_batch"))), cb.get("value"), cb.get("unit"), cb.get("one_core")))
        for kk in ("twitter_json", "array_16mib", "incl_h2d", "on_demand_scan", "full_parse_then_select", "docs_per_s"):
            if kk in v:
                print("    ", kk, v[kk])
