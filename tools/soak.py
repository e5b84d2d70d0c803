#!/usr/bin/env python
"""Soak test of the stage-1 kernel's liveness (experiments only): thousands of launches over random sizes and granule
sizes; every launch must finish with the right count and status 0 (a tripped spin bound would show as status 0x200)."""
import gzip, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simdjson_java_amd as S
doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
maxreps = 320
n0 = len(doc)
buf = torch.zeros(n0 * maxreps + 128, dtype=torch.uint8, device="cuda")
buf[:n0 * maxreps] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda().repeat(maxreps)
out = torch.empty(55263 * maxreps + 16, dtype=torch.int32, device="cuda")
res = torch.zeros(2, dtype=torch.int64, device="cuda")
ctx = S.Context(0, 1 << 20)
work = torch.cuda.Stream(); torch.cuda.synchronize()
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
t0 = time.time()
bad = 0
for it in range(N):
    reps = rng.choice([1, 2, 3, 5, 8, 13, 21, 40, 64, 100, 200, 320]) if rng.random() < 0.7 else rng.randint(1, maxreps)
    ctx.set_tile_steps(rng.choice([0, 0, 1, 2, 4]))
    ctx.stage1_device(buf.data_ptr(), n0 * reps, out.data_ptr(), out.numel(), res.data_ptr(), work.cuda_stream)
    if it % 8 == 7 or it == N - 1:
        torch.cuda.synchronize()
    r = None
    if it % 8 == 7 or it == N - 1:
        r = res.cpu().numpy()
        if int(r[0]) != 55263 * reps or (int(r[1]) & 0xFFFFFFFF) != 0:
            bad += 1
            print("MISMATCH at", it, reps, r)
print("%d launches in %.1f s, %d bad" % (N, time.time() - t0, bad))
