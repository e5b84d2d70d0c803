"""GPU batched mode (BASELINE.json configs[3]/[4]): many documents in one buffer, one stage-1 launch, per-document
split, per-document host stage 2.  Parity: every document's indexes / tree equal the oracle's for that document."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import load_fixture

pytestmark = pytest.mark.gpu


def _pack(docs):
    """NDJSON-style packing: each document followed by one '\\n'; offsets[k] = start of document k."""
    offs = [0]
    buf = bytearray()
    for d in docs:
        buf += d + b"\n"
        offs.append(len(buf))
    return bytes(buf), np.array(offs, dtype=np.uint64)


def _small_docs(rng, n):
    def value(d):
        r = rng.random()
        if d > 3 or r < 0.4:
            return rng.choice(['"s"', '"a\\nb"', '"é€"', '"\\u00e9"', "1", "-2.5e3", "true", "false", "null", '""', "12345678"])
        if r < 0.7:
            return "[" + ",".join(value(d + 1) for _ in range(rng.randint(0, 5))) + "]"
        return "{" + ",".join('"k%d":%s' % (i, value(d + 1)) for i in range(rng.randint(0, 5))) + "}"
    return [value(0).encode() for _ in range(n)]


def test_stage1_batch_matches_per_document_oracle():
    import simdjson_java_amd as S
    rng = random.Random(77)
    docs = _small_docs(rng, 3000) + [load_fixture("github_events.json").rstrip(), b"[]", b"{}", b"7"]
    buf, offs = _pack(docs)
    ctx = S.Context(0, len(buf) + 64)
    try:
        idx, io, st = ctx.stage1_batch(buf, offs)
        assert st == 0
        for k, d in enumerate(docs):
            want, wst = O.stage1(d)
            got = idx[int(io[k]):int(io[k + 1])].astype(np.int64) - int(offs[k])
            assert wst == 0 and np.array_equal(got, want.astype(np.int64)), k
    finally:
        ctx.close()


def test_stage1_batch_isolated_attributes_errors_to_documents():
    """Isolated mode: every document gets the status it would get alone and failing documents do not disturb their
    neighbours -- including two unclosed strings that would cancel in a whole-batch parity, broken UTF-8 next to good
    documents, a long document (several 4 KiB steps), backslash runs across block boundaries, and an empty document."""
    import simdjson_java_amd as S
    rng = random.Random(79)
    docs = _small_docs(rng, 2000)
    bad = [b'["abc', b'{"k": "v}', b'"', b'["' + b"\\" * 101 + b'"]', b'["x' + b"\\" * 70 + b'"',
           bytes([0x5B, 0x22, 0xC3, 0x22, 0x5D]), bytes([0x22, 0x80, 0x22]), b'["a\nb"]', b'["tab\there"]',
           b'["' + b"\xe2\x82" + b'"]', b'["abc']
    for b in bad:
        docs.insert(rng.randrange(len(docs)), b)
    docs.insert(100, load_fixture("github_events.json").rstrip())
    docs.insert(1000, b'["' + b"x" * 9000 + b'", "unclosed')
    docs.insert(1500, b"")
    docs.insert(1600, b'["' + b"y" * 5000 + b'\\\\' * 40 + b'", 1, 2]')
    buf, offs = _pack(docs)
    ctx = S.Context(0, len(buf) + 64)
    try:
        idx, io, ds, st = ctx.stage1_batch_isolated(buf, offs)
        want_or = 0
        nbad = 0
        for k, d in enumerate(docs):
            want, wst = O.stage1(d + b"\n")  # the document's range includes its separator
            assert int(ds[k]) == wst, (k, d[:40], int(ds[k]), wst)
            got = idx[int(io[k]):int(io[k + 1])].astype(np.int64) - int(offs[k])
            if wst:
                assert got.size == 0
                nbad += 1
            else:
                assert np.array_equal(got, want.astype(np.int64)), k
            want_or |= wst
        assert st == want_or and nbad >= 10
        assert int(io[-1]) == idx.size
    finally:
        ctx.close()


def test_parse_batch_trees_and_errors():
    import simdjson_java_amd as S
    rng = random.Random(78)
    docs = _small_docs(rng, 1500)
    # grammar-invalid (but stage-1-valid) documents must fail alone, with the oracle's own error
    bad = [b"[1 1]", b"[1,,1]", b'{"a" 1}', b"[1,2", b'{"a":1,}', b"tru", b"[01]", b'["\\q"]', b'["\\uD800"]', b"1 2", b"[-]"]
    # stage-1-invalid documents too: unclosed strings (two of them: they cancel in a whole-batch parity), broken UTF-8,
    # an unescaped control character -- each must fail alone with the reference's stage-1 message
    bad += [b'["abc', b'{"k": "v', bytes([0x5B, 0x22, 0xC3, 0x22, 0x5D]), b'["a\x01b"]']
    for b in bad:
        docs.insert(rng.randrange(len(docs)), b)
    # root-level strings in front of dropped documents (their closing quote is the document's, not the batch's)
    docs[40:40] = [b'"tail \\t string"', b'"', b'"x"', b'["abc', b'"y"', bytes([0x22, 0x80, 0x22]), b"", b'"z"']
    docs += [b'"end"', b'"']
    buf, offs = _pack(docs)
    p = S.SimdJsonParser(capacity=len(buf) + 64)
    try:
        tapes, strings, errors = p.parse_batch(buf, offs)
        n_bad = 0
        for k, d in enumerate(docs):
            want = O.parse(d + b"\n")
            assert int(errors[k]) == want.error, (k, d, int(errors[k]), want.error)
            if want.error:
                n_bad += 1
                continue
            got = O.Parsed(tapes[k], strings, 0, 0, 0)
            assert got.to_python() == want.to_python(), (k, d)
        assert n_bad >= len(bad)
    finally:
        p.close()


def test_twitter_batch_end_to_end(twitter):
    """configs[4] scaled: twitter.json x 8 as a batch of 8 documents -> 8 trees equal to the oracle's tree."""
    import simdjson_java_amd as S
    reps = 8
    docs = [twitter.rstrip()] * reps
    buf, offs = _pack(docs)
    want = O.parse(twitter).to_python()
    p = S.SimdJsonParser(capacity=len(buf) + 64)
    try:
        tapes, strings, errors = p.parse_batch(buf, offs)
        assert not errors.any()
        for k in range(reps):
            assert O.Parsed(tapes[k], strings, 0, 0, 0).to_python() == want
    finally:
        p.close()


def test_unescape_batch_document_string_offsets():
    """sjmi_unescape_batch: doc_string_offsets[k] is where document k's first string record lies in the batch's string
    buffer = the sum of the record sizes of all earlier documents (stage-1-failing documents contribute none), for
    documents whose structurals straddle the 64-structural groups and the measure kernel's tiles."""
    import simdjson_java_amd as S
    rng = random.Random(80)
    docs = _small_docs(rng, 6000)
    docs.insert(10, load_fixture("github_events.json").rstrip())
    docs.insert(3000, load_fixture("twitter.json").rstrip())
    docs.insert(20, b'["unclosed')
    docs.insert(4000, b"")
    docs.insert(4001, b"")
    docs.append(b'["last \\u00e9 \\n"]')
    # a root-level string is its document's last structural: when the documents behind it are dropped (or empty), its
    # closing quote must still be found inside its own document -- dropped documents ending in a quote included
    docs[30:30] = [b'"tail \\n string"', b'"', b'"plain tail"', bytes([0x22, 0x80, 0x22]), b'["abc', b"", b'"next"', b'"a"', b'"']
    docs += [b'"end"', b'["unclosed at the very end', b'"']
    buf, offs = _pack(docs)
    ctx = S.Context(0, len(buf) + 64)
    try:
        idx, io, ds, st = ctx.stage1_batch_isolated(buf, offs)
        sb, dso, fei, fec = ctx.unescape_batch(len(buf) + 4 * idx.size + 64, len(docs))
        want_sb, want_offs, feo, _ = O.unescape_all(buf + b"\0" * 64, idx)
        assert feo < 0 and fei is None and sb == want_sb
        assert int(dso[0]) == 0 and int(dso[-1]) == len(sb)
        cursor = 0
        for k, d in enumerate(docs):
            assert int(dso[k]) == cursor, k
            if ds[k]:
                continue
            one_sb, _, _, _ = O.unescape_all(d + b"\0" * 64, O.stage1(d + b"\n")[0])
            assert sb[cursor:cursor + len(one_sb)] == one_sb, k
            cursor += len(one_sb)
        assert cursor == len(sb)
        # a plain sjmi_unescape after the batch call knows the documents too
        sb1, fei1, _ = ctx.unescape(len(buf) + 4 * idx.size + 64)
        assert fei1 is None and sb1 == want_sb
        # device-resident form
        import torch
        d_buf = torch.zeros(len(buf) + 128, dtype=torch.uint8, device="cuda")
        d_buf[:len(buf)] = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
        d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
        d_idx = torch.zeros(len(buf) + 2, dtype=torch.int32, device="cuda")
        d_io = torch.zeros(len(docs) + 1, dtype=torch.int64, device="cuda")
        d_ds = torch.zeros(len(docs), dtype=torch.int32, device="cuda")
        d_res = torch.zeros(2, dtype=torch.int64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        ctx.stage1_batch_isolated_device(d_buf.data_ptr(), len(buf), d_offs.data_ptr(), len(docs), d_idx.data_ptr(), d_idx.numel(),
                                         d_io.data_ptr(), d_ds.data_ptr(), d_res.data_ptr(), stream)
        torch.cuda.synchronize()
        count = int(d_res[0].item())
        assert count == idx.size
        d_sb = torch.zeros(len(buf) + 4 * count + 64, dtype=torch.uint8, device="cuda")
        d_dso = torch.zeros(len(docs) + 1, dtype=torch.int64, device="cuda")
        d_ures = torch.zeros(3, dtype=torch.int64, device="cuda")
        ctx.unescape_batch_device(d_buf.data_ptr(), len(buf), d_idx.data_ptr(), count, d_offs.data_ptr(), d_io.data_ptr(),
                                  len(docs), d_sb.data_ptr(), d_sb.numel(), d_dso.data_ptr(), d_ures.data_ptr(), stream)
        torch.cuda.synchronize()
        ures = d_ures.cpu().numpy()
        assert int(ures[0]) == len(want_sb) and int(ures[1]) == 0
        assert bytes(d_sb[:len(want_sb)].cpu().numpy()) == want_sb
        assert np.array_equal(d_dso.cpu().numpy().astype(np.uint64), dso)
    finally:
        ctx.close()


def test_parse_batch_on_many_host_threads(monkeypatch):
    """The host stage 2 of a batch runs on several threads (document ranges balanced by structural count) and the batch
    goes through the GPU as a pipeline of sub-batches on two contexts: trees, errors and tape offsets must not depend on
    the thread count or the number of sub-batches, including failing documents at range boundaries."""
    import simdjson_java_amd as S
    rng = random.Random(81)
    docs = _small_docs(rng, 30000)
    bad = [b"[1 1]", b'{"a" 1}', b"[1,2", b'["\\q"]', b'["abc', bytes([0x5B, 0x22, 0xC3, 0x22, 0x5D]), b"", b"[-]", b'"']
    for i in range(200):
        docs.insert(rng.randrange(len(docs)), bad[i % len(bad)])
    docs.insert(7, load_fixture("github_events.json").rstrip())
    buf, offs = _pack(docs)
    results = {}
    for threads, pipeline in (("1", "1"), ("7", "1"), ("64", "3"), ("5", "2"), ("3", "64")):
        monkeypatch.setenv("SJMI_PARSE_THREADS", threads)
        monkeypatch.setenv("SJMI_PARSE_PIPELINE", pipeline)
        p = S.SimdJsonParser(capacity=len(buf) + 64)
        try:
            results[(threads, pipeline)] = p.parse_batch(buf, offs)
            if pipeline == "2":  # the parser's second batch reuses pool, slabs and contexts
                again = p.parse_batch(buf, offs)
                assert again[1] == results[(threads, pipeline)][1]
                assert all((a is None and b is None) or np.array_equal(a, b) for a, b in zip(again[0], results[(threads, pipeline)][0]))
        finally:
            p.close()
    tapes1, strings1, errors1 = results[("1", "1")]
    assert int((errors1 != 0).sum()) >= 200
    sample = rng.sample(range(len(docs)), 600) + [7]
    want = {}
    for k in sample:
        w = O.parse(docs[k] + b"\n")
        assert int(errors1[k]) == w.error, (k, docs[k])
        want[k] = None if w.error else w.to_python()
    for (threads, pipeline), (tapes, strings, errors) in results.items():
        assert np.array_equal(errors, errors1), (threads, pipeline)
        if pipeline == "1":  # same string-buffer layout: the tapes are equal word for word
            assert strings == strings1
            for a, b in zip(tapes, tapes1):
                assert (a is None and b is None) or np.array_equal(a, b)
        for k in sample:
            if want[k] is not None:
                assert O.Parsed(tapes[k], strings, 0, 0, 0).to_python() == want[k], (threads, pipeline, k)


def test_parse_batch_odd_shapes(monkeypatch):
    """Sub-batch cuts with nothing in them: an empty batch, a single document, one huge document among small ones (some
    sub-batches get no document at all), only failing documents."""
    import simdjson_java_amd as S
    big = load_fixture("twitter.json").rstrip()
    shapes = [[], [b"[1]"], [b"1", big, b'"s"', b"[2]"], [big, b"{}"], [b"[", b'"', b"nul"], [b""] * 5]
    for pipeline in ("1", "3", "8"):
        monkeypatch.setenv("SJMI_PARSE_PIPELINE", pipeline)
        monkeypatch.setenv("SJMI_PARSE_THREADS", "4")
        p = S.SimdJsonParser(capacity=2 * len(big))
        try:
            for docs in shapes:
                buf, offs = _pack(docs)
                tapes, strings, errors = p.parse_batch(buf, offs)
                assert len(tapes) == len(docs)
                for k, d in enumerate(docs):
                    want = O.parse(d + b"\n")
                    assert int(errors[k]) == want.error, (pipeline, k, d[:20])
                    if not want.error:
                        assert O.Parsed(tapes[k], strings, 0, 0, 0).to_python() == want.to_python(), (pipeline, k)
        finally:
            p.close()
