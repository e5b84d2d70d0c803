"""BASELINE.json configs[2]/[3] at test scale: synthetic JSON (50 % strings, 10 % escapes, non-ASCII) repeated on the
device, stage 1 + UTF-8 validation checked against the oracle's closed form; a batch of ~1 KB documents sharded the
way the multi-GPU path shards them."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


def test_synthetic_tile_repeated_on_device():
    import torch
    import simdjson_java_amd as S
    import synth
    tile = synth.synth_tile(target_bytes=1 << 20)
    assert O.utf8_strict(tile)
    idx0, st0 = O.stage1(tile)
    assert st0 == 0 and O.parse(tile).error == 0  # the generator emits valid JSON
    inside = sum(len(s) for s in tile.split(b'"')[1::2])
    assert 0.35 < inside / len(tile) < 0.65  # ~half of the bytes are inside string literals
    reps, n0 = 96, len(tile)
    n = n0 * reps
    work = torch.cuda.Stream()
    with torch.cuda.stream(work):
        buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
        buf[:n] = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(reps)
        cap = idx0.size * reps + 1
        out = torch.empty(cap, dtype=torch.int32, device="cuda")
        res = torch.zeros(2, dtype=torch.int64, device="cuda")
        ctx = S.Context(0, 1 << 20)
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), work.cuda_stream)
        work.synchronize()
        r = res.cpu().numpy()
        assert int(r[0]) == idx0.size * reps and (int(r[1]) & 0xFFFFFFFF) == 0
        want = (torch.from_numpy(idx0.astype(np.int64)).cuda()[None, :] + (torch.arange(reps, device="cuda") * n0)[:, None]).flatten()
        assert torch.equal(out[:idx0.size * reps].to(torch.int64) & 0xFFFFFFFF, want)
        # the same data with one byte broken in the middle must flip exactly the UTF-8 verdict
        buf[n // 2] = 0xFF
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), work.cuda_stream)
        work.synchronize()
        assert (int(res.cpu().numpy()[1]) & 1) == 1
        ctx.close()


def test_small_document_batch_sharded_like_multi_gpu():
    import simdjson_java_amd as S
    from simdjson_java_amd import sharding
    import synth
    docs = synth.small_docs(n=600)
    offs = np.cumsum([0] + [len(d) + 1 for d in docs]).astype(np.uint64)
    total_struct = 0
    for lo, hi in sharding.partition_documents(offs, 4):  # 4 "virtual ranks" on the one GPU
        buf = b"".join(d + b"\n" for d in docs[lo:hi])
        local_offs = (offs[lo:hi + 1] - offs[lo]).astype(np.uint64)
        p = S.SimdJsonParser(capacity=len(buf) + 64)
        try:
            tapes, strings, errors = p.parse_batch(buf, local_offs)
            assert not errors.any()
            for k, d in enumerate(docs[lo:hi]):
                want = O.parse(d)
                assert O.Parsed(tapes[k], strings, 0, 0, 0).to_python() == want.to_python()
                total_struct += want.n_structurals
        finally:
            p.close()
    assert total_struct == sum(O.parse(d).n_structurals for d in docs)
