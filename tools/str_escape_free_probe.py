#!/usr/bin/env python
"""What do the escape sequences cost k_strings?  twitter.json x1024 as it is, and with every backslash replaced by 'x' (no escape in
any string: the wave-uniform shortcuts of the pass -- `any_esc`, `do_u` -- are taken by every granule), both through
sjmi_stage1_device + sjmi_unescape_device; kernel time by HIP events around 20 launches.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import simdjson_java_amd as S
from tools import workloads as W
dev = torch.device("cuda", 0)
work = torch.cuda.Stream(device=dev)  # (stream handle 0 would mean "the context's own stream" to the C ABI)
torch.cuda.set_stream(work)
doc = W.load_twitter()
for name, d in (("twitter x1024", doc), ("twitter x1024, no backslash", doc.replace(b"\\", b"x"))):
    buf, n = W.repeat_on_device(d, 1024, dev)
    ctx = S.Context(device=0, capacity=1 << 20)
    cap = n // 4 + 64
    out = torch.empty(cap, dtype=torch.int32, device=dev)
    res = torch.zeros(2, dtype=torch.int64, device=dev)
    st = work.cuda_stream
    ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st)
    torch.cuda.synchronize()
    count = int(res[0].item())
    sb = torch.empty(n + 4 * count + 64, dtype=torch.uint8, device=dev)
    ures = torch.zeros(3, dtype=torch.int64, device=dev)
    run = lambda: ctx.unescape_device(buf.data_ptr(), n, out.data_ptr(), count, sb.data_ptr(), sb.numel(), ures.data_ptr(), st)
    for _ in range(20):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("%-30s %.4f ms per string pass (%d structurals, %d record bytes)" % (name, e0.elapsed_time(e1) / 20, count, int(ures[0].item())))
    ctx.close()
