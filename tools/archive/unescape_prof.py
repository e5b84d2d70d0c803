#!/usr/bin/env python
"""Runs stage 1 + the string-unescape kernels on twitter x reps (for rocprofv3 --kernel-trace --stats)."""
import gzip, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import simdjson_java_amd as S
import simdjson_java_amd.binding as B
if os.environ.get('SJMI_LIB'):
    B._LIB = os.environ['SJMI_LIB']
doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = len(doc) * reps
work = torch.cuda.Stream(); st = work.cuda_stream
buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
buf[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda().repeat(reps)
cap = 55263 * reps + 1
out = torch.empty(cap, dtype=torch.int32, device="cuda")
res = torch.zeros(2, dtype=torch.int64, device="cuda")
ctx = S.Context(0, 1 << 20)
torch.cuda.synchronize()
ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
sb_cap = n + 4 * cap + 64
sb = torch.empty(sb_cap, dtype=torch.uint8, device="cuda")
ures = torch.zeros(3, dtype=torch.int64, device="cuda")
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    ctx.unescape_device(buf.data_ptr(), n, out.data_ptr(), 55263 * reps, sb.data_ptr(), sb_cap, ures.data_ptr(), st)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    ctx.unescape_device(buf.data_ptr(), n, out.data_ptr(), 55263 * reps, sb.data_ptr(), sb_cap, ures.data_ptr(), st)
torch.cuda.synchronize()
print('unescape total %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3), ures.cpu().numpy())
