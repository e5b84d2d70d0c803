#!/usr/bin/env python
"""A/B of sjmi_parse_document (all stages on the GPU) through the binding: SJMI_LIB selects the library."""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simdjson_java_amd as S
import simdjson_java_amd.binding as B
if os.environ.get('SJMI_LIB'):
    B._LIB = os.environ['SJMI_LIB']
doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
big = b"[" + b",".join(b'{"id":%d,"name":"user %d","tags":["a","b"],"score":%d.5,"ok":true}' % (i, i, i % 97) for i in range(225000)) + b"]"
ctx = S.Context(0, len(big) + 64)
for name, d, n in (("twitter", doc, 300), ("16MiB", big, 20)):
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.parse_document(d)
        t = (time.perf_counter() - t0) / n * 1e3
    print(name, "%.4f ms" % t)
