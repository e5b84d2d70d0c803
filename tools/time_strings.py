"""kernel time of the string pass on twitter x1024 and on a 200k-document slice of configs[3] (no result checks: for ablation
builds, -DSJMI_STR_ABL=1|2|4 = headers | copy loop | flush off; SJMI_LIB selects the build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import simdjson_java_amd as S
import workloads as W
dev = torch.device("cuda", 0)
work = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(work)  # (an explicit stream: handle 0 would be the engine's own)
doc = W.load_twitter()
out = []
for name in ("twitter_x1024", "docs_200k"):
    if name == "twitter_x1024":
        buf, n = W.repeat_on_device(doc, 1024, dev)
    else:
        data, offs = W.unique_docs(0, 200000)
        n = int(offs[-1])
        buf = torch.zeros(n + 128, dtype=torch.uint8, device=dev)
        buf[:n] = torch.from_numpy(data).to(dev)
    cap = n // 3 + 16
    idx = torch.empty(cap, dtype=torch.int32, device=dev)
    res = torch.zeros(2, dtype=torch.int64, device=dev)
    ctx = S.Context(0, 1 << 20)
    st = work.cuda_stream
    ctx.stage1_device(buf.data_ptr(), n, idx.data_ptr(), cap, res.data_ptr(), st)
    torch.cuda.synchronize()
    count = int(res[0].item())
    sb_cap = n + 4 * count + 64
    sb = torch.empty(sb_cap, dtype=torch.uint8, device=dev)
    ures = torch.zeros(3, dtype=torch.int64, device=dev)
    f = lambda: ctx.unescape_device(buf.data_ptr(), n, idx.data_ptr(), count, sb.data_ptr(), sb_cap, ures.data_ptr(), st)
    for _ in range(30):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        f()
    e1.record()
    torch.cuda.synchronize()
    out.append("%s %.4f ms (%.0f GB/s)" % (name, e0.elapsed_time(e1) / 40, n / (e0.elapsed_time(e1) / 40) / 1e6))
    ctx.close()
print(os.environ.get("SJMI_LIB", "base").split("libsjmi_")[-1], " | ".join(out))
