"""does the string pass tolerate writing its records straight into page-locked host memory (zero-copy) for one twitter.json?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import simdjson_java_amd as S
import workloads as W
dev = torch.device("cuda", 0)
work = torch.cuda.Stream(device=dev); torch.cuda.set_stream(work)
doc = W.load_twitter()
buf, n = W.repeat_on_device(doc, 1, dev)
cap = n // 3 + 16
idx = torch.empty(cap, dtype=torch.int32, device=dev)
idx_h = torch.empty(cap, dtype=torch.int32).pin_memory()
res = torch.zeros(2, dtype=torch.int64, device=dev)
ctx = S.Context(0, 1 << 20)
st = work.cuda_stream
sb_cap = n + 4 * cap + 64
sb_d = torch.empty(sb_cap, dtype=torch.uint8, device=dev)
sb_h = torch.empty(sb_cap, dtype=torch.uint8).pin_memory()
ures = torch.zeros(3, dtype=torch.int64, device=dev)
def run(idx_t, sb_t, copy):
    ctx.stage1_device(buf.data_ptr(), n, idx_t.data_ptr(), cap, res.data_ptr(), st)
    ctx.unescape_device(buf.data_ptr(), n, idx_t.data_ptr(), 55263, sb_t.data_ptr(), sb_cap, ures.data_ptr(), st)
    if copy:
        idx_h[:55264].copy_(idx[:55264], non_blocking=True)
        sb_h[:440313].copy_(sb_d[:440313], non_blocking=True)
    work.synchronize()
for name, a, b, c in (("device outputs + D2H", idx, sb_d, True), ("zero-copy indexes + strings", idx_h, sb_h, False), ("zero-copy indexes only (+D2H strings)", idx_h, sb_d, True)):
    for _ in range(20): run(a, b, c)
    t0 = time.perf_counter()
    for _ in range(300): run(a, b, c)
    print("%-40s %.1f us per call" % (name, (time.perf_counter() - t0) / 300 * 1e6))
want = sb_d[:440313].cpu()
run(idx_h, sb_h, False)
print("zero-copy records equal:", bool(torch.equal(sb_h[:440313], want)), "indexes equal:", bool(torch.equal(idx_h[:55263], idx[:55263].cpu())))
