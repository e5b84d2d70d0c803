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


def test_parse_batch_trees_and_errors():
    import simdjson_java_amd as S
    rng = random.Random(78)
    docs = _small_docs(rng, 1500)
    # grammar-invalid (but stage-1-valid) documents must fail alone, with the oracle's own error
    bad = [b"[1 1]", b"[1,,1]", b'{"a" 1}', b"[1,2", b'{"a":1,}', b"tru", b"[01]", b'["\\q"]', b'["\\uD800"]', b"1 2", b"[-]"]
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
