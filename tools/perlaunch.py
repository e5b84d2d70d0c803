#!/usr/bin/env python
"""Per-launch kernel time of stage 1 over a long run (experiments only): clock / power behaviour."""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simdjson_java_amd as S
import simdjson_java_amd.binding as B
if os.environ.get('SJMI_LIB'):
    B._LIB = os.environ['SJMI_LIB']
doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
reps = 1024
n = len(doc) * reps
buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
buf[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda().repeat(reps)
cap = 55263 * reps + 1
out = torch.empty(cap, dtype=torch.int32, device="cuda")
res = torch.zeros(2, dtype=torch.int64, device="cuda")
ctx = S.Context(0, 1 << 20)
work = torch.cuda.Stream(); torch.cuda.synchronize()
st = work.cuda_stream
gap = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
ts = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    ctx.set_profiling(True)
    ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
    torch.cuda.synchronize()
    ms, k = ctx.kernel_time()
    ctx.set_profiling(False)
    ts.append(ms / k)
    if gap: time.sleep(gap)
print("sync each launch, gap %.3f s:" % gap, " ".join("%.0f" % (t * 1e3) for t in ts))
# back-to-back without sync
for rep in range(3):
    ctx.set_profiling(True)
    for i in range(20):
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
    torch.cuda.synchronize()
    ms, k = ctx.kernel_time()
    ctx.set_profiling(False)
    print("20 back-to-back: avg %.1f us" % (ms / k * 1e3))
