"""The configs[3] batch queued N times WITHOUT any check of the result (ablation variants of a kernel produce wrong tapes): for
rocprofv3 --kernel-trace --stats (tools/abl_tok.sh).  SJMI_LIB selects the library."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import importlib
S = importlib.import_module("simdjson-java_amd".replace("-", "_")) if False else __import__("simdjson_java_amd")
from tools import workloads as W
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ctx = S.Context(device=0, capacity=1 << 20)
shard, offs = bench.make_batch_shard(torch, S, W, dev, ctx, 0, n_docs)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    shard.step(st)
torch.cuda.synchronize()
if os.environ.get("SJMI_DBG_AFTER_WARMUP"):  # stage-1 ablation flags (stage1.h DBG_*), set once the index array holds a valid step's indexes
    ctx.debug_set_flags(int(os.environ["SJMI_DBG_AFTER_WARMUP"], 0))
t = time.perf_counter()
for _ in range(steps):
    shard.step(st)
torch.cuda.synchronize()
print("ms per step %.3f" % ((time.perf_counter() - t) / steps * 1e3))
