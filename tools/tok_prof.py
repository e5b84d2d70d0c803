"""Where a wave of k_tok_walk spends its cycles (library built with -DSJMI_TOK_PROF, tools/build_variant.sh): the configs[3] batch,
unchecked.  SJMI_LIB selects the library."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import simdjson_java_amd as S
from tools import workloads as W
lib = ctypes.CDLL(os.environ["SJMI_LIB"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ctx = S.Context(device=0, capacity=1 << 20)
shard, offs = bench.make_batch_shard(torch, S, W, dev, ctx, 0, 1000000)
st = torch.cuda.current_stream().cuda_stream
out = (ctypes.c_ulonglong * 16)()
for _ in range(2):
    shard.step(st)
torch.cuda.synchronize()
lib.sjmi_debug_tok_prof(out, 1)
N = 5
for _ in range(N):
    shard.step(st)
torch.cuda.synchronize()
lib.sjmi_debug_tok_prof(out, 0)
v = [out[i] / N for i in range(16)]
names = ["-", "prologue", "ingest", "step: ring..scan", "step: containers+grammar", "step: words+carries", "flush literals", "between documents"]
tot = v[8]
print("waves %d  cycles per wave %.0f" % (v[9], tot / max(v[9], 1)))
for i in range(1, 8):
    print("%-28s %5.1f %%" % (names[i], 100.0 * v[i] / tot))
print("%-28s %5.1f %%" % ("unaccounted", 100.0 * (tot - sum(v[1:8])) / tot))
