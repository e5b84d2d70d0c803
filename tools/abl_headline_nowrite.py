"""What the index emission costs the headline kernel: twitter.json x reps, stage 1 on the device with and without DBG_NO_WRITE (stage1.h:
the expansion and the index stores skipped; counts and status still right), alternating.  usage: abl_headline_nowrite.py [reps=6801]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import simdjson_java_amd as S
import workloads as W
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6801
dev = torch.device("cuda", 0)
tile = W.load_twitter()
buf, n = W.repeat_on_device(tile, reps, dev)
out = torch.empty(55263 * reps + 16, dtype=torch.int32, device=dev)
res = torch.zeros(2, dtype=torch.int64, device=dev)
ctx = S.Context(0, 1 << 20)
st = torch.cuda.current_stream().cuda_stream
def run(flags, k=40):
    ctx.debug_set_flags(flags)
    for _ in range(5): ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), out.numel(), res.data_ptr(), st)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), out.numel(), res.data_ptr(), st)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
for f in (0, 1, 0, 1):
    ms = run(f)
    print("flags %d: %.4f ms per launch = %.1f GB/s of input" % (f, ms, n / ms / 1e6))
ctx.debug_set_flags(0)
