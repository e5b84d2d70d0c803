"""SimdJsonParser.parse (all stages on the GPU) of twitter.json and of arrays of small objects, wall time per parse through the
Python binding (the same overhead for every library variant: SJMI_LIB selects one) -- for A/B of the single-document chain."""
import gzip, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simdjson_java_amd as S
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
twitter = gzip.open(os.path.join(root, "tests/golden/data/twitter.json.gz")).read()


def array_of(n):
    return b"[" + b",".join(b'{"id":%d,"name":"user %d","tags":["a","b"],"score":%d.5,"ok":true}' % (i, i, i % 97) for i in range(n)) + b"]"


for name, doc, reps in (("twitter.json", twitter, 400), ("array 2000 (0.13 MiB)", array_of(2000), 400), ("array 14000 (0.9 MiB)", array_of(14000), 200),
                        ("array 56000 (3.7 MiB)", array_of(56000), 60), ("array 225000 (15 MiB)", array_of(225000), 20)):
    p = S.SimdJsonParser(capacity=len(doc) + 64, gpu_walk=True)
    for _ in range(10):
        p.parse(doc)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            p.parse(doc)
        best = min(best, (time.perf_counter() - t0) / reps)
    print("%-24s %8.1f us per parse" % (name, best * 1e6))
    p.close()
