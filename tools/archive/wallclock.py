#!/usr/bin/env python
"""Wall-clock time per stage1_device call without profiling events (experiments only)."""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simdjson_java_amd as S
import simdjson_java_amd.binding as B
if os.environ.get('SJMI_LIB'):
    B._LIB = os.environ['SJMI_LIB']
doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
for reps in (1, 1024):
    n = len(doc) * reps
    buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
    buf[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda().repeat(reps)
    out = torch.empty(55263 * reps + 16, dtype=torch.int32, device="cuda")
    res = torch.zeros(2, dtype=torch.int64, device="cuda")
    ctx = S.Context(0, 1 << 20)
    work = torch.cuda.Stream(); torch.cuda.synchronize()
    for _ in range(400):
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), out.numel(), res.data_ptr(), work.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), out.numel(), res.data_ptr(), work.cuda_stream)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 300
    assert int(res.cpu()[0]) == 55263 * reps
    print("x%d: %.1f us per call" % (reps, t * 1e6))
    ctx.close()
