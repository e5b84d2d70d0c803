#!/usr/bin/env python
"""Performance ablation of k_stage1 (experiments only): times the kernel with parts switched off."""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simdjson_java_amd as S

doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = len(doc) * reps
buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
buf[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda().repeat(reps)
cap = 55263 * reps + 1
out = torch.empty(max(cap, n // 8) + (1 << 20), dtype=torch.int32, device="cuda")
res = torch.zeros(2, dtype=torch.int64, device="cuda")
ctx = S.Context(0, 1 << 20)
work = torch.cuda.Stream(); work.wait_stream(torch.cuda.current_stream()); torch.cuda.synchronize()
st = work.cuda_stream
# reference: plain device copy bandwidth
dst = torch.empty_like(buf)
for _ in range(3): dst.copy_(buf)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): dst.copy_(buf)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
print("torch copy: %.3f ms  read %.0f GB/s (r+w %.0f GB/s)" % (t * 1e3, n / t / 1e9, 2 * n / t / 1e9))
del dst
for steps in (1, 2, 4):
    ctx.set_tile_steps(steps)
    for flags, name in [(0, "fast (scanner, static)"), (0x100, "safe (ticket, self look-back)"), (2, "no_lookback"), (1, "no_write"), (3, "no_lookback+no_write")]:
        ctx.debug_set_flags(flags)
        for _ in range(3):
            ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), out.numel(), res.data_ptr(), st)
        ctx.set_profiling(True)
        for _ in range(10):
            ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), out.numel(), res.data_ptr(), st)
        torch.cuda.synchronize()
        ms, k = ctx.kernel_time()
        ctx.set_profiling(False)
        print("steps=%d %-30s %.4f ms  %.0f GB/s" % (steps, name, ms / k, n / (ms / k) / 1e6))
ctx.debug_set_flags(0)
