#!/usr/bin/env python
"""Stage-1 kernel time on structurally dense / sparse documents (experiments only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import simdjson_java_amd as S
from oracle import oracle as O
ctx = S.Context(0, 1 << 20)
work = torch.cuda.Stream(); torch.cuda.synchronize()
def run(name, unit, reps):
    doc = unit * reps
    n = len(doc)
    buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
    buf[:n] = torch.frombuffer(bytearray(unit), dtype=torch.uint8).cuda().repeat(reps)
    idx0, st0 = O.stage1(unit * 4)
    per = idx0.size // 4
    out = torch.empty(per * reps + 1024, dtype=torch.int32, device="cuda")
    res = torch.zeros(2, dtype=torch.int64, device="cuda")
    for _ in range(30):
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), out.numel(), res.data_ptr(), work.cuda_stream)
    ctx.set_profiling(True)
    for _ in range(30):
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), out.numel(), res.data_ptr(), work.cuda_stream)
    torch.cuda.synchronize()
    ms, k = ctx.kernel_time(); ctx.set_profiling(False)
    r = res.cpu().numpy()
    print("%-28s %6.0f MB  %7.1f structurals/KB  count %s  %.3f ms -> %5.0f GB/s in, %5.0f GB/s in+out" % (
        name, n / 1e6, per * reps / n * 1024, "ok" if int(r[0]) == per * reps else "BAD(%d vs %d)" % (int(r[0]), per * reps), ms / k,
        n / (ms / k) / 1e6, (n + 4 * per * reps) / (ms / k) / 1e6))
run("zeros array '0,' (dense)", b"0," * 32768, 4096)
run("small ints '12,'", b"12," * 21845, 4096)
run("short strings", b'"ab",' * 13107, 4096)
run("long strings (1 KB each)", (b'"' + b"x" * 1020 + b'",') * 64, 4096)
run("whitespace only", b" " * 65536, 4096)
