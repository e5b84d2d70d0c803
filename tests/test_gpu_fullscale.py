"""BASELINE.json configs at FULL size on the GPU, checked through properties that do not need the oracle to run over
gigabytes: closed forms over repeated tiles (index[k*S + j] = k*N0 + index0[j], compared on the device), one pool of
documents checked against the oracle word for word and every repetition of the pool checked against the first one.

  configs[1] / north-star  twitter.json x 6801 = 4,294,933,515 B ("4 GiB concatenated twitter.json"), stage 1
  configs[2]               4 GiB synthetic (50 % strings, 10 % escapes, non-ASCII), stage 1 + UTF-8
  configs[3]               ~1 KB documents as an isolated batch: >= 128,000 documents, stage 1 -> strings -> GPU walk
  configs[4]               twitter.json x 1024 as a batch of 1024 documents -> 1024 trees, 86 users each
                           (BenchmarkCorrectnessTest.java:19-42), walked with the JsonValue accessors of the C ABI"""
import os
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


def _stage1_closed_form(tile, reps):
    import torch
    import simdjson_java_amd as S
    import workloads as W
    idx0, st0 = O.stage1(tile)
    assert st0 == 0
    dev = torch.device("cuda", 0)
    work = torch.cuda.Stream()
    with torch.cuda.stream(work):
        buf, n = W.repeat_on_device(tile, reps, dev)
        assert n < (1 << 32)
        cap = idx0.size * reps + 1
        out = torch.empty(cap, dtype=torch.int32, device=dev)
        res = torch.zeros(2, dtype=torch.int64, device=dev)
        ctx = S.Context(0, 1 << 20)
        try:
            ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), work.cuda_stream)
            work.synchronize()
            r = res.cpu().numpy()
            assert int(r[0]) == idx0.size * reps and (int(r[1]) & 0xFFFFFFFF) == 0, r
            ok, bad = W.closed_form_ok(out, idx0, len(tile), reps)
            assert ok, "indexes differ from the oracle's closed form in copies %d.." % bad
            assert int(out[idx0.size * reps].item()) == 0  # BitIndexes.finish sentinel
            # one broken byte near the end flips exactly the UTF-8 verdict, the indexes in front of it stay
            buf[n - 7] = 0xFF
            ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), work.cuda_stream)
            work.synchronize()
            assert (int(res.cpu().numpy()[1]) & 0xFF) & O.ST_UTF8
            ok, _ = W.closed_form_ok(out, idx0, len(tile), reps - 1)
            assert ok
        finally:
            ctx.close()
    return n


def test_twitter_4gib_stage1_closed_form(twitter):
    """north_star: '4 GiB concatenated twitter.json ... bit-identical structural indices': 375,843,663 indexes."""
    import workloads as W
    n = _stage1_closed_form(twitter, W.TWITTER_4G_REPS)
    assert n == 4294933515


def test_synthetic_4gib_stage1_closed_form():
    """configs[2]: the 4 MiB synthetic tile (tools/synth.py) repeated to just under 4 GiB."""
    import workloads as W
    tile = W.synth_tile()
    assert O.utf8_strict(tile)
    reps = (1 << 32) // len(tile) - 1
    n = _stage1_closed_form(tile, reps)
    assert n > 4_200_000_000


def test_twitter_x1024_as_1024_documents_gives_1024_trees(twitter):
    """configs[4]: GPU isolated stage 1 + GPU string records + host stage 2 for a batch of 1024 documents (646 MB).
    Document 0's tape and strings equal the oracle's word for word (as a tree); every other document's tape equals
    document 0's except for the STRING payloads, which are shifted by the offset of its own records, whose bytes equal
    document 0's; and every tree yields 86 users through JsonValue.get / arrayIterator / asBoolean / asString."""
    import simdjson_java_amd as S
    reps = 1024
    doc = twitter.rstrip() + b"\n"
    buf = doc * reps
    offs = (np.arange(reps + 1, dtype=np.uint64) * np.uint64(len(doc)))
    want = O.parse(twitter)
    assert want.error == 0
    p = S.SimdJsonParser(capacity=len(buf) + 64)
    try:
        tapes, strings, errors = p.parse_batch(buf, offs)
        assert not errors.any() and len(tapes) == reps
        assert O.Parsed(tapes[0], strings, 0, 0, 0).to_python() == want.to_python()
        t0 = tapes[0]
        is_str = (t0 >> np.uint64(56)) == np.uint64(ord('"'))
        # (the second word of a number is raw payload: exclude positions that follow an 'l' / 'd' word)
        ty = (t0 >> np.uint64(56)).astype(np.uint8)
        num2 = np.zeros(t0.size, dtype=bool)
        i = 0
        while i < t0.size:
            if ty[i] in (ord("l"), ord("d")):
                num2[i + 1] = True
                i += 2
            else:
                i += 1
        is_str &= ~num2
        rec0 = int(t0[is_str][0]) & 0x00FFFFFFFFFFFFFF
        s_len = len(want.strings)
        base_bytes = strings[rec0:rec0 + s_len]
        assert base_bytes == want.strings
        for k in range(1, reps):
            tk = tapes[k]
            assert tk.size == t0.size, k
            assert np.array_equal(tk[~is_str], t0[~is_str]), k
            shift = tk[is_str].astype(np.int64) - t0[is_str].astype(np.int64)
            assert (shift == shift[0]).all(), k
            reck = rec0 + int(shift[0])
            assert strings[reck:reck + s_len] == base_bytes, k
        # BenchmarkCorrectnessTest.countUniqueTwitterUsersWithDefaultProfile on every tree, through the C-ABI JsonValue
        for k in range(reps):
            users = set()
            for tweet in p.batch_root(k).get("statuses").arrayIterator():
                user = tweet.get("user")
                if user.get("default_profile").asBoolean():
                    users.add(user.get("screen_name").asString())
            assert len(users) == 86, k
    finally:
        p.close()


def test_128k_document_batch_full_device_pipeline():
    """configs[3] at >= 100,000 documents: 32 repetitions of a pool of 4,000 unique ~1 KB documents (128,000 documents,
    ~126 MB), isolated stage 1 -> string records -> GPU walk, all device-resident.  Repetition 0 is checked against
    the oracle document by document (indexes, tree); every other repetition against repetition 0 on the device."""
    import torch
    import simdjson_java_amd as S
    import workloads as W
    docs, unit, lens = W.small_doc_pool(4000)
    reps = 32
    n_docs, n = len(docs) * reps, len(unit) * reps
    offs = W.batch_offsets(lens, reps)
    dev = torch.device("cuda", 0)
    buf, _ = W.repeat_on_device(unit, reps, dev)
    d_offs = torch.from_numpy(offs.view(np.int64)).to(dev)
    cap = n // 3 + 1
    d_idx = torch.empty(cap, dtype=torch.int32, device=dev)
    d_io = torch.zeros(n_docs + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n_docs, dtype=torch.int32, device=dev)
    res = torch.zeros(2, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ctx = S.Context(0, 1 << 20)
    try:
        ctx.stage1_batch_isolated_device(buf.data_ptr(), n, d_offs.data_ptr(), n_docs, d_idx.data_ptr(), cap, d_io.data_ptr(),
                                         d_st.data_ptr(), res.data_ptr(), st)
        torch.cuda.synchronize()
        count = int(res[0].item())
        assert (int(res[1].item()) & 0xFFFFFFFF) == 0 and not bool(d_st.any().item())
        d_sb = torch.zeros(n + 4 * count + 64, dtype=torch.uint8, device=dev)
        d_dso = torch.zeros(n_docs + 1, dtype=torch.int64, device=dev)
        d_ures = torch.zeros(3, dtype=torch.int64, device=dev)
        ctx.unescape_batch_device(buf.data_ptr(), n, d_idx.data_ptr(), count, d_offs.data_ptr(), d_io.data_ptr(), n_docs,
                                  d_sb.data_ptr(), d_sb.numel(), d_dso.data_ptr(), d_ures.data_ptr(), st)
        tcap = 2 * count + 2 * n_docs + 8
        d_tape = torch.zeros(tcap, dtype=torch.int64, device=dev)
        d_to = torch.zeros(n_docs + 1, dtype=torch.int64, device=dev)
        d_err = torch.zeros(n_docs, dtype=torch.int32, device=dev)
        d_wres = torch.zeros(4, dtype=torch.int64, device=dev)
        ctx.walk_batch_device(buf.data_ptr(), d_offs.data_ptr(), n_docs, d_idx.data_ptr(), count, d_io.data_ptr(), d_st.data_ptr(),
                              d_sb.data_ptr(), d_dso.data_ptr(), 0, 1024, d_tape.data_ptr(), tcap, d_to.data_ptr(),
                              d_err.data_ptr(), d_wres.data_ptr(), st)
        torch.cuda.synchronize()
        assert int(d_ures[1].item()) == 0 and not bool(d_err.any().item())
        # ---- repetition 0 against the oracle, document by document ----
        m = len(docs)
        io = d_io[:m + 1].cpu().numpy()
        to = d_to[:m + 1].cpu().numpy()
        s0, t0n = int(io[m]), int(to[m])
        idx0 = d_idx[:s0].cpu().numpy().view(np.uint32)
        tape0 = d_tape[:t0n].cpu().numpy().view(np.uint64)
        sb_len0 = int(d_dso[m].item())
        sb0 = bytes(d_sb[:sb_len0].cpu().numpy())
        for k, d in enumerate(docs):
            w_idx, w_st = O.stage1(d + b"\n")
            assert w_st == 0
            got = idx0[int(io[k]):int(io[k + 1])].astype(np.int64) - int(offs[k])
            assert np.array_equal(got, w_idx.astype(np.int64)), k
            want = O.parse(d + b"\n")
            assert want.error == 0
            assert O.Parsed(tape0[int(to[k]):int(to[k + 1])], sb0, 0, 0, 0).to_python() == want.to_python(), k
        # ---- every repetition against repetition 0, on the device ----
        assert count == s0 * reps and int(d_wres[0].item()) == t0n * reps and int(d_ures[0].item()) == sb_len0 * reps
        rep_io = d_io[:n_docs].view(reps, m)
        assert torch.equal(rep_io - rep_io[:, :1], rep_io[:1].expand(reps, m) - rep_io[0, 0])
        got_idx = (d_idx[:count].to(torch.int64) & 0xFFFFFFFF).view(reps, s0)
        assert torch.equal(got_idx - (torch.arange(reps, device=dev) * len(unit))[:, None], got_idx[:1].expand(reps, s0))
        sbv = d_sb[:sb_len0 * reps].view(reps, sb_len0)
        assert torch.equal(sbv, sbv[:1].expand(reps, sb_len0))
        tp = d_tape[:t0n * reps].view(reps, t0n)
        is_str = torch.from_numpy(((tape0 >> np.uint64(56)) == np.uint64(ord('"')))).to(dev)
        # second words of numbers hold raw payload (they could look like a string word): mask them out
        ty = (tape0 >> np.uint64(56)).astype(np.uint8)
        num2 = np.zeros(t0n, dtype=bool)
        i = 0
        while i < t0n:
            if ty[i] in (ord("l"), ord("d")):
                num2[i + 1] = True
                i += 2
            else:
                i += 1
        is_str &= ~torch.from_numpy(num2).to(dev)
        shift = torch.where(is_str[None, :], (torch.arange(reps, device=dev) * sb_len0)[:, None], torch.zeros((), dtype=torch.int64, device=dev))
        assert torch.equal(tp - shift, tp[:1].expand(reps, t0n))
    finally:
        ctx.close()


def test_one_million_unique_documents_every_document_against_the_oracle():
    """configs[3] at full size, EVERY document (round 5: the bench checked a 10,000-document sample, the suite 128,000): the
    1,000,000 unique ~1 KB documents of tools/docgen.c through sjmi_parse_batch_device_optimistic; per document the digest of its
    tape (every word, STRING words by their record's bytes), the hash of its structural indexes, its count and its verdict
    against the oracle's own per-document parse (oracle/sj_oracle.c sjo_digest_many / sjo_digest_outputs)."""
    import sys
    import torch
    import simdjson_java_amd as S
    from tests.conftest import ROOT
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import workloads as W
    W.build_docgen()
    dev = torch.device("cuda", 0)
    ctx = S.Context(0, 1 << 20)
    try:
        shard, offs = bench.make_batch_shard(torch, S, W, dev, ctx, 0, 1000000)
        shard.step(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        c = shard.check()
        assert c["failed_documents"] == 0 and c["host_documents"] == 0 and getattr(shard, "rejected_steps", 0) == 0
        assert bench.check_batch_all(torch, O, shard, offs) == 1000000
    finally:
        ctx.close()
