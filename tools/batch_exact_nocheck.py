"""The configs[3] batch with ONE document failing stage 1, queued N times through the EXACT entry (sjmi_parse_batch_device) without
checks: for rocprofv3 --kernel-trace --stats (tools/prof_kernels.sh) -- what a rejected batch costs, kernel by kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import simdjson_java_amd as S
from tools import workloads as W
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "bad"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ctx = S.Context(device=0, capacity=1 << 20)
shard, offs = bench.make_batch_shard(torch, S, W, dev, ctx, 0, n_docs)
entry = sys.argv[4] if len(sys.argv) > 4 else "exact"   # exact | rejected
if mode == "bad":
    shard.buf[int(offs[n_docs // 2 + 1]) - 4] = 0xFF
elif mode == "nosep":
    import numpy as np
    shard.buf[torch.from_numpy(np.asarray(offs[1:], dtype=np.int64) - 1).to(dev)] = 0x20
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    shard.step(st, exact=(entry == 'exact'), rejected=(entry == 'rejected'))
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    shard.step(st, exact=(entry == 'exact'), rejected=(entry == 'rejected'))
torch.cuda.synchronize()
print("ms per step %.3f" % ((time.perf_counter() - t) / steps * 1e3), shard.check())
