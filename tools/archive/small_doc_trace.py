#!/usr/bin/env python
"""the drop-in call on a SMALL document (sjmi_parser_parse, host walker placement), timed from C++, for a kernel trace"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import simdjson_java_amd as S
import ondemand_bench
L = ondemand_bench.load_bench_lib()
n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 14
d = b"[" + b",".join(b'{"id":%d,"name":"user %d","tags":["a","b"],"score":%d.5,"ok":true}' % (i, i, i % 97) for i in range(n_items)) + b"]"
buf = (C.c_uint8 * len(d)).from_buffer_copy(d)
for mode in (False, True):
    p = S.SimdJsonParser(capacity=len(d) + 64, gpu_walk=mode)
    secs, a, b = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    for rep in range(3):
        L.odb_run(p._h, buf, len(d), 3, 300, C.byref(secs), C.byref(a), C.byref(b))
    print("gpu_walk", mode, len(d), "B: %.4f ms" % (secs.value / 300 * 1e3))
    p.close()
