#!/usr/bin/env python
"""Per-step cost of the cooperative walker on single documents of different content (one wave sweeps the document):
time of sjmi_parse_document minus the same without the walk is not separable from outside, so the whole call is timed
for documents with the same number of structurals."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simdjson_java_amd as S
n = 27000
docs = {
    "short ints": b"[" + b",".join(b"1" for _ in range(n)) + b"]",
    "18-digit ints": b"[" + b",".join(b"505874924095815681" for _ in range(n)) + b"]",
    "floats": b"[" + b",".join(b"-12.5e3" for _ in range(n)) + b"]",
    "short strings": b"[" + b",".join(b'"a"' for _ in range(n)) + b"]",
    "atoms": b"[" + b",".join(b"true" for _ in range(n)) + b"]",
    "empty arrays": b"[" + b",".join(b"[]" for _ in range(n // 2)) + b"]",
    "nested pairs": b"[" + b",".join(b"[1]" for _ in range(n // 2)) + b"]",
    "objects": b"[" + b",".join(b'{"a":1}' for _ in range(n // 3)) + b"]",
}
ctx = S.Context(0, 4 << 20)
for name, d in docs.items():
    for _ in range(3):
        tape, sb, err, st = ctx.parse_document(d)
    assert err == 0
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.parse_document(d)
    t = (time.perf_counter() - t0) / 10
    idx, _ = ctx.stage1(d)
    steps = (idx.size + 63) // 64
    print("%-14s %7d structurals %5d steps: %.3f ms per call = %.2f us per step" % (name, idx.size, steps, t * 1e3, t * 1e6 / steps))
