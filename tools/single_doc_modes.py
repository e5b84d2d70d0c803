"""Single-document latency of the two walker placements (host walker over the stage-1 output vs all three stages on the
device), twitter.json and a 64 MiB array of small objects; SJMI_COOP_CHUNKS=0 forces the single-wave sweep."""
import gzip
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simdjson_java_amd as S  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
twitter = gzip.open(os.path.join(root, "tests/golden/data/twitter.json.gz")).read()
big = b"[" + b",".join(b'{"id":%d,"name":"user %d","tags":["a","b"],"score":%d.5,"ok":true}' % (i, i, i % 97) for i in range(900000)) + b"]"
def array_of(n):
    return b"[" + b",".join(b'{"id":%d,"name":"user %d","tags":["a","b"],"score":%d.5,"ok":true}' % (i, i, i % 97) for i in range(n)) + b"]"


cases = [("twitter.json", twitter, 200)]
for n, reps in ((14000, 100), (56000, 40), (225000, 15)):
    d = array_of(n)
    cases.append(("array of %dk objects (%.1f MiB)" % (n // 1000, len(d) / 2 ** 20), d, reps))
cases.append(("array of 900k objects (%.0f MiB)" % (len(big) / 2 ** 20), big, 5))
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
