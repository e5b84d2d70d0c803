"""world_size-2 CPU (gloo) test of the multi-GPU batched path's host logic: document partitioning + the single
all_gather of per-shard counts.  Per-shard counts come from the oracle here (no GPU in this container); on the GPU
box the same functions run over "nccl" with counts produced by the kernels (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _docs():
    import random
    rng = random.Random(5)
    docs = []
    for i in range(400):
        n = rng.randint(1, 40)
        docs.append(("[" + ",".join('{"k":"v%d","n":%d}' % (j, j) for j in range(n)) + "]").encode())
    return docs


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import simdjson_java_amd  # noqa: F401  (package import shim)
    from simdjson_java_amd import sharding
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    docs = _docs()
    offs = np.cumsum([0] + [len(d) + 1 for d in docs]).astype(np.uint64)
    lo, hi = sharding.partition_documents(offs, world)[rank]
    structurals = strings = failed = 0
    for d in docs[lo:hi]:
        p = O.parse(d)
        structurals += p.n_structurals
        strings += len(p.strings)
        failed += int(p.error != 0)
    g = sharding.gather_counts([hi - lo, structurals, strings, failed])
    off = sharding.global_offsets(g)
    q.put((rank, g.tolist(), off.tolist(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_contiguous_and_balanced():
    sys.path.insert(0, ROOT)
    import simdjson_java_amd  # noqa: F401
    from simdjson_java_amd import sharding
    docs = _docs()
    offs = np.cumsum([0] + [len(d) + 1 for d in docs]).astype(np.uint64)
    for world in (1, 2, 3, 8):
        parts = sharding.partition_documents(offs, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(docs)
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        sizes = [int(offs[b] - offs[a]) for a, b in parts]
        assert max(sizes) - min(sizes) <= 2 * max(len(d) + 1 for d in docs)


def test_two_rank_gather_of_counts():
    import torch.multiprocessing as mp
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    docs = _docs()
    total_struct = sum(O.parse(d).n_structurals for d in docs)
    (r0, g0, off0, rng0), (r1, g1, off1, rng1) = res
    assert g0 == g1, "every rank must see the same gathered table"
    assert g0[0][0] + g0[1][0] == len(docs) and g0[0][1] + g0[1][1] == total_struct
    assert off0[0] == [0, 0, 0, 0] and off0[1] == g0[0]
    assert rng0[1] == rng1[0]
