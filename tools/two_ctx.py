#!/usr/bin/env python
"""Two contexts launching stage 1 concurrently on two streams (experiments only): does anybody trip a spin bound?"""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simdjson_java_amd as S
doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = len(doc) * reps
bufs, outs, ress, ctxs, streams = [], [], [], [], []
for k in range(2):
    b = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
    b[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda().repeat(reps)
    bufs.append(b)
    outs.append(torch.empty(55263 * reps + 16, dtype=torch.int32, device="cuda"))
    ress.append(torch.zeros(2, dtype=torch.int64, device="cuda"))
    ctxs.append(S.Context(0, 1 << 20))
    ctxs[-1].debug_set_flags(int(sys.argv[2], 0) if len(sys.argv) > 2 else 0)
    streams.append(torch.cuda.Stream())
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(20):
    for k in range(2):
        ctxs[k].stage1_device(bufs[k].data_ptr(), n, outs[k].data_ptr(), outs[k].numel(), ress[k].data_ptr(), streams[k].cuda_stream)
torch.cuda.synchronize()
t = time.perf_counter() - t0
for k in range(2):
    r = ress[k].cpu().numpy()
    print("ctx %d: count ok %s status 0x%x" % (k, int(r[0]) == 55263 * reps, int(r[1]) & 0xFFFFFFFF))
print("40 launches of %d MB in %.3f s -> %.0f GB/s aggregate" % (n // 1000000, t, 40 * n / t / 1e9))
