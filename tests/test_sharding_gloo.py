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


class _StubEngine:
    """Stands in for binding.Context in the CPU test: the same parse_batch_device signature, but the per-shard result
    record comes from the oracle (the kernels need a GPU; what is under test is the sharding / gather logic around them)."""

    def parse_batch_device(self, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets,
                           d_doc_status, d_sb, sb_capacity, d_doc_string_offsets, max_depth, d_tape, tape_capacity,
                           d_tape_offsets, d_doc_errors, d_result, stream=0):
        import ctypes as C
        from oracle import oracle as O
        buf = (C.c_uint8 * total_len).from_address(d_buf)
        offs = (C.c_int64 * (n_docs + 1)).from_address(d_doc_offsets)
        res = (C.c_int64 * 9).from_address(d_result)
        raw = bytes(buf)
        structurals = strings = words = failed = 0
        for k in range(n_docs):
            p = O.parse(raw[offs[k]:offs[k + 1]])
            if p.error:
                failed += 1
                continue
            structurals += p.n_structurals
            strings += len(p.strings)
            words += p.tape.size
        for i, v in enumerate([structurals, 0, strings, 0, 0, words, 0, failed, 0]):
            res[i] = v


def _shard_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import simdjson_java_amd  # noqa: F401
    from simdjson_java_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    docs = _docs()
    docs[7] = b"[1 1]"      # one broken document in rank 0's share
    docs[-3] = b'{"a":1,}'  # and one in the last rank's
    offs = np.cumsum([0] + [len(d) + 1 for d in docs]).astype(np.uint64)
    lo, hi = sharding.partition_documents(offs, world)[rank]
    data = b"".join(d + b"\n" for d in docs[lo:hi])
    shard = sharding.BatchShard(_StubEngine(), data, offs[lo:hi + 1] - offs[lo], "cpu")
    g = sharding.sharded_step(shard)
    q.put((rank, g.tolist(), shard.check()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_step():
    """bench.py --gpus N's step on two gloo ranks: partition -> BatchShard.step (engine stubbed) -> count gather."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    docs = _docs()
    docs[7] = b"[1 1]"
    docs[-3] = b'{"a":1,}'
    good = [O.parse(d) for d in docs]
    (r0, g0, c0), (r1, g1, c1) = res
    assert g0 == g1
    assert g0[0][0] + g0[1][0] == len(docs)
    assert g0[0][1] + g0[1][1] == sum(p.n_structurals for p in good if not p.error)
    assert g0[0][2] + g0[1][2] == sum(len(p.strings) for p in good if not p.error)
    assert g0[0][3] == 1 and g0[1][3] == 1
    assert c0["failed_documents"] == 1 and c0["documents"] == g0[0][0]


class _StubShardEngine:
    """binding.Context.stage1_shard_device for the CPU test: the per-byte form of stage 1 (SURVEY.md 8(a) a3') over the left
    halo (for the escape / previous-scalar carries only) and then over the shard, entered with the given parity."""

    def stage1_shard_device(self, d_buf, length, halo, is_last, entry_parity, d_indexes, cap, d_result, stream=0, halo_from_start=False):
        import ctypes as C
        raw = bytes((C.c_uint8 * (halo + length)).from_address(d_buf - halo))
        out = (C.c_uint32 * cap).from_address(d_indexes)
        res = (C.c_int64 * 2).from_address(d_result)
        esc = prev_nqs = 0
        in_str, unesc, n = 0, 0, 0
        for pos, c in enumerate(raw):
            if pos == halo:
                in_str = entry_parity
            escaped = esc
            esc = 1 if (c == 0x5C and not escaped) else 0
            q = c == 0x22 and not escaped
            if q:
                in_str ^= 1
            ws = c in (0x20, 0x09, 0x0A, 0x0D)
            op = c in b",:[]{}\x0c\x1a"
            scalar = not (op or ws)
            start = scalar and not prev_nqs
            prev_nqs = scalar and not q
            if pos >= halo:
                if (op or start) and not (in_str ^ q):
                    out[n] = pos - halo
                    n += 1
                if c <= 0x1F and in_str:
                    unesc = 1
        res[0] = n
        res[1] = (2 if in_str else 0) | (4 if unesc else 0)


def _split_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import simdjson_java_amd  # noqa: F401
    from simdjson_java_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    doc = _split_doc()
    a, b = sharding.split_points(len(doc), world)[rank]
    h = min(64, a)
    shard = sharding.DocumentShard(_StubShardEngine(), doc[a - h:b], h, rank == world - 1, "cpu")
    r = sharding.resolve_split_document(shard, lambda: None)
    idx = shard.idx[:r["count"]].numpy().view(np.uint32).astype(np.int64) + a
    q.put((rank, r, idx.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _split_doc():
    # the boundary of two ranks (byte 640) lies inside the long string: rank 1 must find out that it starts inside one
    return b'{"k":[1,2,3],"s":"' + b"x y z, " * 120 + b'\\" still inside","t":[true,false,null,{"u":"v"}]}   '


def test_two_rank_split_document():
    """sharding.resolve_split_document over gloo: scan, gather the parity flips, re-scan where the true entry parity is 1,
    gather the counts -- rank 1 starts inside a string and has to run twice."""
    import torch.multiprocessing as mp
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    doc = _split_doc()
    want, wst = O.stage1(doc)
    (r0, o0, i0), (r1, o1, i1) = res
    assert o0["entry_parity"] == 0 and o1["entry_parity"] == 1
    assert o0["total"] == o1["total"] == want.size and o1["offset"] == o0["count"]
    assert i0 + i1 == want.tolist()
    assert o0["status"] == o1["status"] == wst == 0


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
