"""Single-document latency of the two walker placements (host walker over the stage-1 output vs all three stages on the
device), twitter.json and a 64 MiB array of small objects; SJMI_COOP_CHUNKS=0 forces the single-wave sweep."""
import gzip
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import simdjson_java_amd as S  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
twitter = gzip.open(os.path.join(root, "tests/golden/data/twitter.json.gz")).read()
big = b"[" + b",".join(b'{"id":%d,"name":"user %d","tags":["a","b"],"score":%d.5,"ok":true}' % (i, i, i % 97) for i in range(900000)) + b"]"
def array_of(n):
    return b"[" + b",".join(b'{"id":%d,"name":"user %d","tags":["a","b"],"score":%d.5,"ok":true}' % (i, i, i % 97) for i in range(n)) + b"]"


cases = [("twitter.json", twitter, 200)]
for n, reps in ((10, 300), (200, 300), (2000, 300), (14000, 100), (56000, 40), (225000, 15)):
    d = array_of(n)
    cases.append(("array of %d objects (%.3f MiB)" % (n, len(d) / 2 ** 20), d, reps))
cases.append(("array of 900k objects (%.0f MiB)" % (len(big) / 2 ** 20), big, 5))
import ctypes as C  # noqa: E402
import ondemand_bench  # noqa: E402
L = ondemand_bench.load_bench_lib()
for name, doc, reps in cases:
    buf = (C.c_uint8 * len(doc)).from_buffer_copy(doc)
    for mode in (False, True):  # the same call without the Python binding's copies (tools/ondemand_bench.cpp, mode 3)
        p = S.SimdJsonParser(capacity=len(doc) + 64, gpu_walk=mode)
        secs, a, b = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        best = 1e9
        for rep in range(3):
            assert L.odb_run(p._h, buf, len(doc), 3, reps, C.byref(secs), C.byref(a), C.byref(b)) == 0
            best = min(best, secs.value / reps)
        print("%s gpu_walk=%s: sjmi_parser_parse from C++ %.3f ms = %.2f GB/s" % (name, mode, best * 1e3, len(doc) / best / 1e9))
        p.close()
for name, doc, reps in cases:
    for mode in (False, True):
        p = S.SimdJsonParser(capacity=len(doc) + 64, gpu_walk=mode)
        for _ in range(3):
            p.parse(doc)
        t0 = time.perf_counter()
        for _ in range(reps):
            p.parse(doc)
        t = (time.perf_counter() - t0) / reps
        print("%s gpu_walk=%s chunks=%s: parse %.3f ms = %.2f GB/s" % (name, mode, os.environ.get("SJMI_COOP_CHUNKS", "1"), t * 1e3, len(doc) / t / 1e9))
        p.close()
