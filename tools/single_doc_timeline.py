"""one sjmi_parser_parse(twitter.json) with all stages on the GPU, as a timeline (run under rocprofv3 --kernel-trace --memory-copy-trace)"""
import ctypes as C, gzip, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import simdjson_java_amd as S
import ondemand_bench
doc = gzip.open(os.path.join(R, "tests/golden/data/twitter.json.gz")).read()
L = ondemand_bench.load_bench_lib()
buf = (C.c_uint8 * len(doc)).from_buffer_copy(doc)
p = S.SimdJsonParser(capacity=len(doc) + 64, gpu_walk=True)
secs, a, b = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
for rep in range(3):
    assert L.odb_run(p._h, buf, len(doc), 3, 200, C.byref(secs), C.byref(a), C.byref(b)) == 0
    print("twitter.json all-device: %.4f ms per parse" % (secs.value / 200 * 1e3))
p.close()
