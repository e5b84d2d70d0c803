#!/usr/bin/env python
"""sjmi_parser_parse of ONE document from a host buffer, timed from C++ (tools/ondemand_bench.cpp mode 3), both placements of
stage 2, over document sizes from 1 KiB to 64 MiB -- the table of DESIGN.md 4.4.  Run on the GPU box."""
import ctypes as C, gzip, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import simdjson_java_amd as S
import simdjson_java_amd.binding as B
if os.environ.get('SJMI_LIB'):
    B._LIB = os.environ['SJMI_LIB']  # (A/B of library variants: tools/build_variant.sh)
import ondemand_bench
L = ondemand_bench.load_bench_lib()


def arr(n):
    return b"[" + b",".join(b'{"id":%d,"name":"user %d","tags":["a","b"],"score":%d.5,"ok":true}' % (i, i, i % 97) for i in range(n)) + b"]"


docs = [("1 KiB", arr(14)), ("4 KiB", arr(60)), ("14 KiB", arr(200)), ("136 KiB", arr(1900)),
        ("twitter.json", gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()),
        ("1 MiB", arr(14200)), ("3.9 MiB", arr(55000)), ("16 MiB", arr(225000)), ("64 MiB", arr(900000))]
out = {}
for name, d in docs:
    buf = (C.c_uint8 * len(d)).from_buffer_copy(d)
    n = 300 if len(d) < (2 << 20) else (40 if len(d) < (32 << 20) else 10)
    row = {"bytes": len(d)}
    for mode, key in ((False, "host_walker_ms"), (True, "gpu_walker_ms")):
        p = S.SimdJsonParser(capacity=len(d) + 64, gpu_walk=mode)
        secs, a, b = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        ms = 1e9
        for rep in range(4):
            if L.odb_run(p._h, buf, len(d), 3, n, C.byref(secs), C.byref(a), C.byref(b)):
                raise SystemExit("sjmi_parser_parse failed")
            if rep:
                ms = min(ms, secs.value / n * 1e3)
        row[key] = round(ms, 4)
        p.close()
    out[name] = row
    print(name, row, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "single_doc_sizes.json"), "w"), indent=1)
