"""SimdJsonParser.parse of twitter.json with all three stages on the GPU, N times: for rocprofv3 --kernel-trace (the timeline of one
parse: which kernels, how long, the gaps between them).  tools/prof_kernels.sh single python tools/single_doc_trace.py"""
import gzip, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simdjson_java_amd as S
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
doc = gzip.open(os.path.join(root, "tests/golden/data/twitter.json.gz")).read()
p = S.SimdJsonParser(capacity=len(doc) + 64, gpu_walk=True)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    p.parse(doc)
p.close()
