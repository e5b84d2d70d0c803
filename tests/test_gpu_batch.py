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
