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
torch.cuda.synchronize()
t0 = time.time()
bad = 0
for it in range(launches):
    ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), work.cuda_stream)
    if it % 16 == 15 or it == launches - 1:
        torch.cuda.synchronize()
        r = res.cpu().numpy()
        if int(r[0]) != idx0.size * reps or (int(r[1]) & 0xFFFFFFFF) != 0:
            bad += 1
            print("%s: MISMATCH at launch %d: %r" % (tag, it, r), flush=True)
ok, where = W.closed_form_ok(out, idx0, len(doc), reps)
print("%s: %d launches of %d MB in %.1f s, %d bad, final indexes %s" % (tag, launches, n // 1000000, time.time() - t0, bad, "ok" if ok else "WRONG"), flush=True)
ctx.close()
sys.exit(0 if (bad == 0 and ok) else 1)
