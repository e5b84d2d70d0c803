#!/usr/bin/env python
"""Secondary measurements quoted in DESIGN.md (not the bench line): 4 GiB variant, host-buffer (PCIe) path,
string-unescape kernels, end-to-end parse."""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import simdjson_java_amd as S

doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
n0 = len(doc)
work = torch.cuda.Stream()
st = work.cuda_stream

def device_run(reps, iters=10):
    n = n0 * reps
    buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
    buf[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda().repeat(reps)
    cap = 55263 * reps + 1
    out = torch.empty(cap, dtype=torch.int32, device="cuda")
    res = torch.zeros(2, dtype=torch.int64, device="cuda")
    ctx = S.Context(0, 1 << 20)
    torch.cuda.synchronize()
    for _ in range(2):
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
    torch.cuda.synchronize()
    r = res.cpu().numpy()
    assert int(r[0]) == 55263 * reps and (int(r[1]) & 0xFFFFFFFF) == 0, r
    # spot parity: first and last copy against the closed form
    from oracle import oracle as O
    idx0, _ = O.stage1(doc)
    first = (out[:55263].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
    last = (out[55263 * (reps - 1):55263 * reps].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
    assert np.array_equal(first, idx0.astype(np.int64)) and np.array_equal(last, idx0.astype(np.int64) + n0 * (reps - 1))
    ctx.set_profiling(True)
    for _ in range(iters):
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
    torch.cuda.synchronize()
    ms, k = ctx.kernel_time()
    print("stage1 device-resident: twitter x%d = %d B: %.4f ms/launch -> %.0f GB/s (%.1f%% of 8 TB/s)" % (reps, n, ms / k, n / (ms / k) / 1e6, n / (ms / k) / 1e6 / 80))
    # unescape on the same data
    sb_cap = n + 4 * cap + 64
    sb = torch.empty(sb_cap, dtype=torch.uint8, device="cuda")
    ures = torch.zeros(3, dtype=torch.int64, device="cuda")
    ctx.unescape_device(buf.data_ptr(), n, out.data_ptr(), 55263 * reps, sb.data_ptr(), sb_cap, ures.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(work):
        e0.record()
        for _ in range(5):
            ctx.unescape_device(buf.data_ptr(), n, out.data_ptr(), 55263 * reps, sb.data_ptr(), sb_cap, ures.data_ptr(), st)
        e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    u = ures.cpu().numpy()
    print("unescape (4 launches): %.3f ms for %d string-buffer bytes -> %.0f GB/s of document, first_error_inv=%d" % (t, int(u[0]), n / t / 1e6, int(u[1])))
    ctx.close()

device_run(1024)
if len(sys.argv) > 1 and sys.argv[1] == "4g":
    device_run(6801, iters=5)
# host-buffer path (PCIe H2D + kernel + D2H of the indexes)
reps = 64
hdoc = doc * reps
ctx = S.Context(0, len(hdoc) + 64)
ctx.stage1(hdoc)
t0 = time.perf_counter()
for _ in range(5):
    idx, stt = ctx.stage1(hdoc)
t = (time.perf_counter() - t0) / 5
print("stage1 host-buffer path (pageable numpy in/out, PCIe both ways): twitter x%d = %d B: %.2f ms -> %.1f GB/s" % (reps, len(hdoc), t * 1e3, len(hdoc) / t / 1e9))
ctx.close()
p = S.SimdJsonParser(capacity=len(doc) + 64)
p.parse(doc)
t0 = time.perf_counter()
for _ in range(20):
    p.parse(doc)
t = (time.perf_counter() - t0) / 20
print("SimdJsonParser.parse(twitter.json) end to end (H2D + GPU stage1 + GPU unescape + D2H + host stage 2): %.3f ms = %.0f ops/s" % (t * 1e3, 1 / t))
