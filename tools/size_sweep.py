#!/usr/bin/env python
"""Stage-1 kernel time vs input size and granule size (experiments only): picks the thresholds of stage1_pick_steps."""
import gzip, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simdjson_java_amd as S
doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
ctx = S.Context(0, 1 << 20)
work = torch.cuda.Stream(); torch.cuda.synchronize()
st = work.cuda_stream
for reps in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    n = len(doc) * reps
    buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
    buf[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda().repeat(reps)
    cap = 55263 * reps + 1
    out = torch.empty(cap, dtype=torch.int32, device="cuda")
    res = torch.zeros(2, dtype=torch.int64, device="cuda")
    line = "x%-4d %7.1f MB:" % (reps, n / 1e6)
    for steps in (1, 2, 4):
        ctx.set_tile_steps(steps)
        for _ in range(20):
            ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
        ctx.set_profiling(True)
        for _ in range(50):
            ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
        torch.cuda.synchronize()
        ms, k = ctx.kernel_time()
        ctx.set_profiling(False)
        assert int(res.cpu()[0]) == 55263 * reps
        line += "  S=%d %7.1f us (%5.0f GB/s)" % (steps, ms / k * 1e3, n / (ms / k) / 1e6)
    print(line)
