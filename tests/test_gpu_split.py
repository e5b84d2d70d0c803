"""ONE document split over several ranks / fed as a stream of chunks (sjmi_stage1_shard_device, sharding.DocumentShard /
resolve_split_document / stream_document): the concatenation of the shards' indexes
and the combined verdict must be what stage 1 gives for the whole document -- with shard boundaries inside strings,
behind backslashes, inside numbers and inside UTF-8 sequences.  The ranks are virtual here (one GPU): the collectives of
the protocol are covered by tests/test_sharding_gloo.py."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_gpu_stage1 import _json_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import simdjson_java_amd as S
    c = S.Context(device=0, capacity=1 << 20)
    yield c
    c.close()


def _run_split(ctx, doc, bounds, halo=64):
    """the protocol of resolve_split_document, all ranks on one device, in rank order"""
    import torch
    from simdjson_java_amd import sharding
    dev = torch.device("cuda", 0)
    shards = []
    for r, (a, b) in enumerate(bounds):
        h = min(halo, a) // 64 * 64
        shards.append(sharding.DocumentShard(ctx, doc[a - h:b], h, r == len(bounds) - 1, dev, halo_from_start=(h == a)))
    for s in shards:
        s.run(0)
    torch.cuda.synchronize()
    flips = [s.outcome()[2] for s in shards]
    reran = 0
    for r, s in enumerate(shards):
        entry = sum(flips[:r]) & 1
        if entry:
            s.run(1)
            reran += 1
    torch.cuda.synchronize()
    idx, status, after = [], 0, 0
    for (a, b), s in zip(bounds, shards):
        count, st, after = s.outcome()
        status |= st
        idx.append(s.idx[:count].cpu().numpy().view(np.uint32).astype(np.int64) + a)
    if after:
        status |= O.ST_UNCLOSED
    return (np.concatenate(idx) if idx else np.zeros(0, np.int64)), status, reran


def sharding_mod():
    from simdjson_java_amd import sharding
    return sharding


def _check(ctx, doc, bounds):
    want_idx, want_st = O.stage1(doc)
    got_idx, got_st, reran = _run_split(ctx, doc, bounds)
    assert got_st == want_st, (bounds, got_st, want_st)
    assert np.array_equal(got_idx, want_idx.astype(np.int64)), bounds
    return reran


def test_twitter_split_over_virtual_ranks(ctx, twitter):
    from simdjson_java_amd import sharding
    doc = twitter * 16  # 10 MB; 631,515 is odd: every copy meets the boundaries in another phase
    reran = 0
    for parts in (2, 3, 8):
        reran += _check(ctx, doc, sharding.split_points(len(doc), parts))
    assert reran > 0  # some shard did start inside a string (58 % of twitter.json's bytes are string bytes)


def test_every_boundary_of_small_hazard_documents(ctx):
    """two shards, the boundary at EVERY multiple of 64: strings, escapes ending exactly at the boundary, multi-byte
    characters and numbers straddling it, unclosed strings, control characters inside strings, broken UTF-8"""
    rng = random.Random(41)
    docs = [_json_like(rng, 2048) for _ in range(3)]
    docs.append(b'["' + b"x" * 300 + b'\\\\\\"' + b"y" * 400 + b'", 12345678901234567890, "' + "é€😀".encode() * 120 + b'"]')
    docs.append(b'{"unclosed": "' + b"z" * 900)
    docs.append(b'["a\x01b", "' + b"q" * 700 + b'\x02", 1]')
    docs.append(b'["' + "€".encode() * 200 + b'\xe2\x82", "' + b"r" * 500 + b'"]')  # truncated sequence in the middle
    docs.append(b"\\" * 700 + b'"x" [1,2,3]')  # (a run longer than the 64-byte halo must still be exact up to the halo's reach)
    for d in docs:
        for cut in range(64, len(d) // 64 * 64 + 1, 64):
            if cut >= len(d):
                break
            if d.startswith(b"\\\\") and 64 < cut <= 640:
                # a backslash run that fills the whole halo: the call says so (SJMI_ST_HALO) instead of guessing
                with pytest.raises(sharding_mod().HaloTooShort):
                    _check(ctx, d, [(0, cut), (cut, len(d))])
                continue
            _check(ctx, d, [(0, cut), (cut, len(d))])
    # with a halo that covers the run, long backslash runs are exact everywhere
    d = b"\\" * 700 + b'"x" [1,2,3]'
    want_idx, want_st = O.stage1(d)
    for cut in (64, 128, 640, 704):
        got_idx, got_st, _ = _run_split(ctx, d, [(0, cut), (cut, len(d))], halo=1024)
        assert got_st == want_st and np.array_equal(got_idx, want_idx.astype(np.int64)), cut


def test_many_shards_of_a_fuzzed_document(ctx):
    rng = random.Random(42)
    d = bytearray(_json_like(rng, 300000))
    for _ in range(60):  # lone quotes: the parity changes from shard to shard
        d[rng.randrange(len(d))] = 0x22
    d = bytes(d)
    from simdjson_java_amd import sharding
    for parts in (2, 5, 16, 37):
        _check(ctx, d, sharding.split_points(len(d), parts))


def test_document_stream_mode(ctx, twitter):
    """the same entry point fed chunk by chunk on one GPU (parity carried on the host, no re-run needed)"""
    import torch
    from simdjson_java_amd import sharding
    doc = twitter * 4
    want_idx, want_st = O.stage1(doc)
    cuts = [0, 65536, 65536 * 3, 1 << 20, (1 << 20) + 64, len(doc)]
    chunks = [doc[a:b] for a, b in zip(cuts, cuts[1:])]
    parts, st = sharding.stream_document(ctx, torch.device("cuda", 0), chunks)
    got = np.concatenate([ix.astype(np.int64) + base for base, ix in parts])
    assert st == want_st and np.array_equal(got, want_idx.astype(np.int64))


def test_backslash_run_that_fills_the_halo_is_reported(ctx):
    """a backslash run of 64 or more bytes in front of a shard / chunk boundary hides the escape carry behind
    the halo.  The call must say so (SJMI_ST_HALO) instead of deriving the parity from the visible part; with more halo the
    result is the whole document's, and the stream mode repeats the chunk with more halo by itself."""
    import torch
    from simdjson_java_amd import sharding
    dev = torch.device("cuda", 0)
    for run in (64, 65, 127, 128, 129, 200):
        # the run ends exactly at the boundary (a multiple of 64): an odd run escapes the quote behind it, an even one does not
        pre = b'["' + b"a" * (256 - 2 - run) + b"\\" * run
        assert len(pre) == 256
        doc = pre + b'","x"]       '
        want_idx, want_st = O.stage1(doc)
        # one halo block: the run (>= 64 backslashes) fills it
        sh = sharding.DocumentShard(ctx, doc[192:], 64, True, dev)
        sh.run(1)  # (the boundary lies inside the first string)
        torch.cuda.synchronize()
        with pytest.raises(sharding.HaloTooShort):
            sh.outcome()
        # enough halo to see where the run begins
        sh = sharding.DocumentShard(ctx, doc, 256, True, dev)
        sh.run(1)
        torch.cuda.synchronize()
        count, st, after = sh.outcome()
        first = sharding.DocumentShard(ctx, doc[:256], 0, False, dev)
        first.run(0)
        torch.cuda.synchronize()
        c0, st0, flip0 = first.outcome()
        assert flip0 == 1
        got = np.concatenate([first.idx[:c0].cpu().numpy().view(np.uint32), sh.idx[:count].cpu().numpy().view(np.uint32) + 256])
        assert np.array_equal(got, want_idx), run
        assert (st0 | st | (2 if after else 0)) == want_st, run
        # the stream mode with the default halo of 64 bytes repeats the second chunk with more halo
        parts, status = sharding.stream_document(ctx, dev, [doc[:256], doc[256:]], halo=64)
        got = np.concatenate([idx + base for base, idx in parts])
        assert np.array_equal(got, want_idx) and status == want_st, run


def test_stream_and_split_through_the_c_abi(ctx, twitter):
    """sjmi_stream_* / sjmi_split_*: the protocol's state lives in C (include/sjmi.h), no chunk is scanned twice in stream mode,
    an odd-parity shard exactly twice in split mode."""
    import torch
    dev = torch.device("cuda", 0)
    # ---- stream: twitter x4 in uneven chunks, and the backslash run that fills the default halo ----
    doc = twitter * 4
    want_idx, want_st = O.stage1(doc)
    cuts = [0, 65536, 65536 * 3, 1 << 20, (1 << 20) + 64, len(doc)]
    s = ctx.stream(max(b - a for a, b in zip(cuts, cuts[1:])))
    got, st = [], 0
    for a, b in zip(cuts, cuts[1:]):
        base, idx, st = s.push(doc[a:b], b == len(doc))
        assert base == a
        got.append(idx.astype(np.int64) + base)
    s.close()
    assert st == want_st and np.array_equal(np.concatenate(got), want_idx.astype(np.int64))
    for run in (64, 65, 129, 1000, 1001):
        pre = b'["' + b"a" * (2048 - 2 - run) + b"\\" * run
        d = pre + b'","x"]       ' + b'"unclosed'
        want_idx, want_st = O.stage1(d)
        s = ctx.stream(4096)
        b0, i0, _ = s.push(d[:2048], False)
        b1, i1, st = s.push(d[2048:], True)
        s.close()
        assert st == want_st and np.array_equal(np.concatenate([i0.astype(np.int64), i1.astype(np.int64) + 2048]), want_idx.astype(np.int64)), run
    # a run longer than everything the stream keeps (4 KiB): reported, not guessed
    import simdjson_java_amd as S
    s = ctx.stream(8192)
    s.push(b'["' + b"\\" * 8190, False)
    with pytest.raises(S.SjmiError):
        s.push(b'\\"x"]' + b" " * 59, True)
    s.close()
    # ---- split: twitter x16 over 5 virtual ranks ----
    from simdjson_java_amd import sharding
    doc = twitter * 16
    want_idx, want_st = O.stage1(doc)
    bounds = sharding.split_points(len(doc), 8)
    bufs, idxs, sp = [], [], []
    for r, (a, b) in enumerate(bounds):
        h = min(64, a)
        t = torch.zeros(h + (b - a) + 128, dtype=torch.uint8, device=dev)
        t[:h + b - a] = torch.frombuffer(bytearray(doc[a - h:b]), dtype=torch.uint8).to(dev)
        ix = torch.empty(b - a + 66, dtype=torch.int32, device=dev)
        bufs.append(t)
        idxs.append(ix)
        sp.append(ctx.split(t.data_ptr() + h, b - a, h, h == a, r == len(bounds) - 1, ix.data_ptr(), ix.numel()))
    flips = [x.scan()[0] for x in sp]
    out, status, rescans = [], 0, 0
    for r, x in enumerate(sp):
        entry = sum(flips[:r]) & 1
        rescans += entry
        count, st, after = x.resolve(entry)
        status |= st
        out.append(idxs[r][:count].cpu().numpy().view(np.uint32).astype(np.int64) + bounds[r][0])
        x.close()
    if after:
        status |= O.ST_UNCLOSED
    assert status == want_st and np.array_equal(np.concatenate(out), want_idx.astype(np.int64)), rescans
