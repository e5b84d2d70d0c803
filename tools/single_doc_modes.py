import sys, time, gzip
sys.path.insert(0, '/root/repo')
import simdjson_java_amd as S
doc = gzip.open('/root/repo/tests/golden/data/twitter.json.gz').read()
for mode in (False, True):
    p = S.SimdJsonParser(capacity=len(doc)+64, gpu_walk=mode)
    for _ in range(20): p.parse(doc)
    t0=time.perf_counter()
    for _ in range(200): p.parse(doc)
    t=(time.perf_counter()-t0)/200
    print("gpu_walk=%s: parse(twitter.json) %.3f ms = %.0f ops/s" % (mode, t*1e3, 1/t))
    p.close()
