"""Two configs[3] batches in flight on two streams (two contexts, two shards): does a second batch's stage 1 / string pass fill what
the first one's walker leaves idle?  An experiment (profiles/r5/README.md); run on the GPU box: overlap_probe.py [documents] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import simdjson_java_amd as S
from tools import workloads as W
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
shards, streams = [], []
for i in range(2):
    ctx = S.Context(device=0, capacity=1 << 20)
    shard, offs = bench.make_batch_shard(torch, S, W, dev, ctx, 0, n_docs)
    shards.append(shard)
    streams.append(torch.cuda.Stream(device=dev))
def run(which, n):
    for s in which:
        for _ in range(2):
            shards[s].step(streams[s].cuda_stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        s = which[i % len(which)]
        shards[s].step(streams[s].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
one = run([0], steps)
two = run([0, 1], steps)
bad = [sh.check() for sh in shards]
print("one stream %.3f ms per batch; two streams alternating %.3f ms per batch (%.1f %%)" % (one, two, 100.0 * (one - two) / one))
print([(c["failed_documents"], c["stage1_status"]) for c in bad])
