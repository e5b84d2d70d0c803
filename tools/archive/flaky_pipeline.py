import sys
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import simdjson_java_amd as S
from simdjson_java_amd import sharding
from tests.test_gpu_pipeline import _docs
from tests.test_gpu_batch import _pack
docs=_docs(); buf,offs=_pack(docs)
ctx=S.Context(0,1<<20)
shard=sharding.BatchShard(ctx, buf, offs, torch.device("cuda",0))
st=torch.cuda.current_stream().cuda_stream
prev=None
for it in range(30):
    shard.step(st); torch.cuda.synchronize()
    r=shard.result.cpu().numpy().tolist()
    sb=bytes(shard.sb[:r[2]].cpu().numpy())
    tp=shard.tape[:r[5]].cpu().numpy().tobytes()
    key=(tuple(r), hash(sb), hash(tp))
    if prev is not None and key!=prev: print("DIFF at", it, r, prev[0])
    prev=key
print("done", r)
