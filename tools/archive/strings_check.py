#!/usr/bin/env python
"""GPU check + timing of the streaming string pass (csrc/strings.hip) through the C ABI: the reference files and
twitter x reps against the oracle's string buffer, then the kernel time from HIP events around back-to-back calls."""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import simdjson_java_amd as S
import simdjson_java_amd.binding as B
if os.environ.get('SJMI_LIB'):
    B._LIB = os.environ['SJMI_LIB']
from oracle import oracle as O

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
ctx = S.Context(0, 64 << 20)
for name in ("twitter.json", "github_events.json", "wide_bench.json"):
    d = gzip.open(os.path.join(ROOT, "tests/golden/data/%s.gz" % name)).read()
    idx, st = ctx.stage1(d)
    want, _, feo, _ = O.unescape_all(d + b"\0" * 64, idx)
    got, fei, fec = ctx.unescape(len(d) + 4 * idx.size + 64)
    print(name, "OK" if got == want and fei is None else "MISMATCH", len(got), len(want), fei, fec, flush=True)
    if got != want:
        n = min(len(got), len(want))
        bad = next((i for i in range(n) if got[i] != want[i]), n)
        print("  first difference at", bad, got[max(0, bad - 16):bad + 16], want[max(0, bad - 16):bad + 16])

n = len(doc) * reps
idx0, _ = O.stage1(doc)
want_sb, _, _, _ = O.unescape_all(doc + b"\0" * 64, idx0)
work = torch.cuda.Stream(); st = work.cuda_stream
buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
buf[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda().repeat(reps)
cap = idx0.size * reps + 1
out = torch.empty(cap, dtype=torch.int32, device="cuda")
res = torch.zeros(2, dtype=torch.int64, device="cuda")
dctx = S.Context(0, 1 << 20)
torch.cuda.synchronize()
with torch.cuda.stream(work):
    dctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
    sb_cap = n + 4 * cap + 64
    sb = torch.zeros(sb_cap, dtype=torch.uint8, device="cuda")
    ures = torch.zeros(3, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    dctx.unescape_device(buf.data_ptr(), n, out.data_ptr(), idx0.size * reps, sb.data_ptr(), sb_cap, ures.data_ptr(), st)
    torch.cuda.synchronize()
    u = ures.cpu().numpy()
    print("x%d: total %d (want %d) first_error_inv %d flags/nstrings %x" % (reps, int(u[0]), len(want_sb) * reps, int(u[1]), int(u[2]) & 0xFFFFFFFFFFFFFFFF), flush=True)
    ok = int(u[0]) == len(want_sb) * reps
    if ok:
        sbv = sb[:len(want_sb) * reps].view(reps, len(want_sb))
        w = torch.frombuffer(bytearray(want_sb), dtype=torch.uint8).cuda()
        eq = (sbv == w.unsqueeze(0)).all(dim=1)
        print("copies identical to the oracle's buffer: %d of %d" % (int(eq.sum()), reps), flush=True)
        if not bool(eq.all()):
            r = int((~eq).nonzero()[0])
            diff = (sbv[r] != w).nonzero().flatten()
            print("  copy", r, "first differing bytes at", diff[:8].tolist(), "of", diff.numel())
            b = int(diff[0])
            print("  got ", bytes(sbv[r][max(0, b - 24):b + 24].cpu().numpy().tolist()))
            print("  want", bytes(w[max(0, b - 24):b + 24].cpu().numpy().tolist()))
    for _ in range(30):
        dctx.unescape_device(buf.data_ptr(), n, out.data_ptr(), idx0.size * reps, sb.data_ptr(), sb_cap, ures.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(work)
    K = 30
    for _ in range(K):
        dctx.unescape_device(buf.data_ptr(), n, out.data_ptr(), idx0.size * reps, sb.data_ptr(), sb_cap, ures.data_ptr(), st)
    e1.record(work)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    u = ures.cpu().numpy()
    if int(u[1]):
        v = int(u[1]) & 0xFFFFFFFFFFFFFFFF
        print("spin statistics: flushes %d, of which waited %d, polls %d" % (v >> 44, v & 0xFFFFF, (v >> 20) & 0xFFFFFF))
    print("string pass: %.4f ms per call = %.0f GB/s of document (%d B)" % (ms, n / ms / 1e6, n), flush=True)
    # stage 1 with the parity side output, for comparison with the committed number
    for _ in range(40):
        dctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
    e0.record(work)
    for _ in range(K):
        dctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
    e1.record(work)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    print("stage 1: %.4f ms per call = %.0f GB/s" % (ms, n / ms / 1e6), res.cpu().numpy(), flush=True)
