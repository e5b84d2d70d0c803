#!/usr/bin/env python
"""Where the time of one SimdJsonParser.parse(twitter.json) goes (experiments only)."""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import simdjson_java_amd as S
doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
ctx = S.Context(0, len(doc) + 64)
def t(f, n=200):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
idx, st = ctx.stage1(doc)
print("sjmi_stage1 (H2D + memset + kernel + D2H result + D2H indexes, incl. python/numpy overhead): %.1f us" % t(lambda: ctx.stage1(doc)))
print("sjmi_unescape (3 kernels + D2H strings): %.1f us" % t(lambda: ctx.unescape(len(doc) + 4 * idx.size + 64)))
a = np.frombuffer(doc, dtype=np.uint8); ix = np.empty(a.size + 2, dtype=np.uint32); sbuf = np.empty(a.size * 3 + 64, dtype=np.uint8)
print("sjmi_stage1_unescape (fused: H2D, 4 kernels, 2 syncs, D2H indexes + strings; pageable numpy buffers): %.1f us" % t(lambda: ctx.stage1_unescape(a, ix, sbuf)))
p = S.SimdJsonParser(capacity=len(doc) + 64)
print("parse end to end: %.1f us" % t(lambda: p.parse(doc)))
# device-side only
d = torch.frombuffer(bytearray(doc + b"\0" * 128), dtype=torch.uint8).cuda()
out = torch.empty(idx.size + 16, dtype=torch.int32, device="cuda")
res = torch.zeros(2, dtype=torch.int64, device="cuda")
work = torch.cuda.Stream(); torch.cuda.synchronize()
def dev():
    ctx.stage1_device(d.data_ptr(), len(doc), out.data_ptr(), out.numel(), res.data_ptr(), work.cuda_stream)
    torch.cuda.synchronize()
print("stage1_device + sync (memset + kernel + result copy): %.1f us" % t(dev))
ctx.set_profiling(True)
for _ in range(50): ctx.stage1_device(d.data_ptr(), len(doc), out.data_ptr(), out.numel(), res.data_ptr(), work.cuda_stream)
torch.cuda.synchronize()
ms, k = ctx.kernel_time(); print("k_stage1 on twitter.json (631 KB): %.1f us" % (ms / k * 1e3))
