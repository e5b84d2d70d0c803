#!/usr/bin/env python
"""One of two PROCESSES hammering stage 1 on the same GPU (tests/test_gpu_two_process.py): persistent FAST-mode launches
from two processes compete for residency, which is exactly the situation the liveness design has to survive -- every
launch must either complete with the right indexes or be repeated in SAFE mode (sjmi_set_auto_safe), never hang and
never return something else.  usage: two_proc_worker.py <tag> <launches> <reps>"""
import gzip, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import simdjson_java_amd as S
import workloads as W
from oracle import oracle as O

tag, launches, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
with_strings = len(sys.argv) > 4 and sys.argv[4] == "strings"  # the string pass (another persistent kernel) behind every launch
doc = W.load_twitter()
idx0, _ = O.stage1(doc)
dev = torch.device("cuda", 0)
buf, n = W.repeat_on_device(doc, reps, dev)
cap = idx0.size * reps + 1
out = torch.empty(cap, dtype=torch.int32, device=dev)
res = torch.zeros(2, dtype=torch.int64, device=dev)
ctx = S.Context(0, 1 << 20)
ctx.set_auto_safe(True)
work = torch.cuda.Stream()
if with_strings:
    want_sb, _, _, _ = O.unescape_all(doc + b"\0" * 64, idx0)
    sb_cap = n + 4 * cap + 64
    sb = torch.zeros(sb_cap, dtype=torch.uint8, device=dev)
    ures = torch.zeros(3, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
t0 = time.time()
bad = 0
for it in range(launches):
    ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), work.cuda_stream)
    if with_strings:
        ctx.unescape_device(buf.data_ptr(), n, out.data_ptr(), idx0.size * reps, sb.data_ptr(), sb_cap, ures.data_ptr(), work.cuda_stream)
    if it % 16 == 15 or it == launches - 1:
        torch.cuda.synchronize()
        r = res.cpu().numpy()
        if int(r[0]) != idx0.size * reps or (int(r[1]) & 0xFFFFFFFF) != 0:
            bad += 1
            print("%s: MISMATCH at launch %d: %r" % (tag, it, r), flush=True)
        if with_strings:
            u = ures.cpu().numpy()
            if int(u[0]) != len(want_sb) * reps or int(u[1]) != 0 or (int(u[2]) & 0xF) != 0:
                bad += 1
                print("%s: STRING PASS MISMATCH at launch %d: %r" % (tag, it, u), flush=True)
ok, where = W.closed_form_ok(out, idx0, len(doc), reps)
if with_strings:
    w = torch.frombuffer(bytearray(want_sb), dtype=torch.uint8).to(dev)
    ok = ok and bool((sb[:len(want_sb) * reps].view(reps, len(want_sb)) == w.unsqueeze(0)).all())
print("%s: %d launches of %d MB in %.1f s, %d bad, final indexes %s" % (tag, launches, n // 1000000, time.time() - t0, bad, "ok" if ok else "WRONG"), flush=True)
ctx.close()
sys.exit(0 if (bad == 0 and ok) else 1)
