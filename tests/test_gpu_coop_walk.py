"""The cooperative GPU walker (csrc/coop_walk.hip: stage 2 as scans and local predicates, a wave per document) against
the oracle's sequential walker: error code for error code, tape word for tape word -- single documents of any size
through sjmi_parse_document (all three stages on the device, twitter.json = 864 steps of one wave), and batches through
the same gpu_walk harness as the lane-per-document kernel it replaces."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import load_fixture
from tests.golden import vectors as V
from tests.test_gpu_walk import check_against_oracle, gpu_walk
from tests.test_host_walk import GRAMMAR
from tests.walk_common import NEEDS_HOST

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import simdjson_java_amd as S
    c = S.Context(device=0, capacity=48 * 1024 * 1024)
    yield c
    c.close()


def _adversarial(rng, n):
    """tests/test_coop_walk_model.py's generator: every grammar error of JsonIterator.java:68-193, brackets of the wrong kind,
    missing / doubled separators, broken atoms and numbers, at every depth."""
    def value(d):
        r = rng.random()
        if d > 4 or r < 0.4:
            return rng.choice(['"s"', '"a\\nb"', '"é€"', "1", "-2.5e3", "true", "false", "null", '""', "12345678", "0.000001", "tru", "01",
                               "1.", "", "falsey", "nul", "]", "}", ":", ",", "[", "{", '"\\q"', '"\\ud800"', "1e400", "-0", "[]", "{}"])
        if r < 0.7:
            return "[" + rng.choice([",", ", ", " ,", " "]).join(value(d + 1) for _ in range(rng.randint(0, 5))) + rng.choice(["]", "]", "]", "}", ""])
        return "{" + rng.choice([",", ",", " "]).join('%s%s%s' % (rng.choice(['"k%d"' % i, '"k"', "1", ""]), rng.choice([":", ":", " : ", "", ","]), value(d + 1))
                                                      for i in range(rng.randint(0, 5))) + rng.choice(["}", "}", "}", "]", ""])
    return [value(0).encode() for _ in range(n)]


def _single(ctx, doc, max_depth=1024):
    want = O.parse(doc, max_depth=max_depth)
    tape, strings, err, st = ctx.parse_document(doc, max_depth=max_depth)
    assert st == want.stage1_status
    if err == NEEDS_HOST:
        return "host"
    assert err == want.error, (doc[:80], err, want.error)
    if err == 0:
        assert np.array_equal(tape, want.tape), doc[:80]
        assert strings == want.strings
    return "ok"


@pytest.mark.parametrize("name", ["twitter.json", "github_events.json", "wide_bench.json"])
def test_reference_files_as_single_documents(ctx, name):
    assert _single(ctx, load_fixture(name)) == "ok"
    assert _single(ctx, load_fixture(name).rstrip()) == "ok"


def test_grammar_vectors_single(ctx):
    for doc in GRAMMAR:
        assert _single(ctx, doc) == "ok", doc
    for text, n, msg, cite in V.GRAMMAR:
        d = text.encode()
        assert _single(ctx, d[:n] if n is not None else d) == "ok", cite
    for text, msg, cite in V.STRING_ERRORS:
        assert _single(ctx, text.encode()) == "ok", cite
    for text, want, cite in V.VALID_DOCS:
        assert _single(ctx, text.encode()) == "ok", cite
    for depth in (3, 4, 5, 10):
        for doc in (b"[[[[1]]]]", b'{"a":{"b":{"c":1}}}', b"[[[[]]]]", b"[" * 10 + b"]" * 10, b'[{"a":[{}]}]'):
            assert _single(ctx, doc, max_depth=depth) == "ok"


def test_adversarial_documents_single_and_batched(ctx):
    rng = random.Random(331)
    docs = _adversarial(rng, 3000)
    for d in docs[:700]:
        _single(ctx, d)
    tapes, strings, errors = gpu_walk(ctx, docs)
    check_against_oracle(docs, tapes, strings, errors)
    assert int((errors > 0).sum()) > 1000 and int((errors == 0).sum()) > 300


def test_documents_spanning_many_steps(ctx):
    """Containers that open in one 64-structural step and close many steps later (the per-wave stack in LDS), long arrays
    whose element count accumulates over steps, nesting to the device limit, and the count saturation of
    ArrayParsingTest.java:74-95 at reduced scale."""
    rng = random.Random(332)

    def nested(depth, width):
        if depth == 0:
            return rng.choice(["1", '"x"', "true", "null", "-2.5", "[]", "{}"])
        if rng.random() < 0.5:
            return "[" + ",".join(nested(depth - 1, width) for _ in range(rng.randint(1, width))) + "]"
        return "{" + ",".join('"k%d":%s' % (i, nested(depth - 1, width)) for i in range(rng.randint(1, width))) + "}"
    docs = [nested(6, 4).encode() for _ in range(40)]
    docs += [("[" + ",".join(str(i) for i in range(n)) + "]").encode() for n in (1, 63, 64, 65, 127, 128, 129, 1000, 70000)]
    docs += [("[" * d + "7" + "]" * d).encode() for d in (1, 31, 32, 33, 62, 63)]
    docs += [("[" * d + "]" * d).encode() for d in (1, 2, 63, 64)]
    docs += [b'{"a":' * 40 + b"[1,2,3]" + b"}" * 40, b"[" + b"[1,[2,[3,[4]]]]," * 200 + b"0]"]
    for d in docs:
        assert _single(ctx, d) == "ok", d[:60]
    tapes, strings, errors = gpu_walk(ctx, docs)
    check_against_oracle(docs, tapes, strings, errors)
    # deeper than the 64 levels the walker keeps in registers: levels 64 .. 1023 live in global memory -- the tape is still the
    # oracle's, word for word, up to the reference's default maxDepth (SimdJsonParser.java:7, JsonIterator.java:20-23)
    rng = random.Random(78)
    deep = [b"[" * d + b"1" + b"]" * d for d in (64, 65, 100, 500, 1000, 1022, 1023)]
    deep += [b'{"a":' * d + b"[1,2,{}]" + b"}" * d for d in (64, 200, 1020)]
    deep.append(b"[" * 300 + b",".join(b"[" * 70 + b'"x",2' + b"]" * 70 for _ in range(5)) + b"]" * 300)  # up and down across level 64 .. 370
    kinds = [rng.choice("[{") for _ in range(900)]  # arrays and objects alternating at random, 900 deep
    deep.append(("".join("[" if k == "[" else '{"k":' for k in kinds) + "0" + "".join("]" if k == "[" else "}" for k in reversed(kinds))).encode())
    for d in deep:
        assert _single(ctx, d) == "ok", (len(d), d[:20])
    tapes, strings, errors = gpu_walk(ctx, deep)
    import sys
    limit = sys.getrecursionlimit()
    sys.setrecursionlimit(20000)  # (the tree comparison of the checker recurses per level)
    try:
        check_against_oracle(deep, tapes, strings, errors)
    finally:
        sys.setrecursionlimit(limit)
    # commas and counts of containers that live in the overflow levels; errors there
    for d in (b"[" * 100 + b"1,2,3" + b"]" * 100, b"[" * 100 + b"1 2" + b"]" * 100, b"[" * 100 + b"1" + b"]" * 99, b"[" * 99 + b"1" + b"]" * 100,
              b'{"a":' * 80 + b'"v", "b":1' + b"}" * 80, b"[" * 80 + b"[],[]" + b"]" * 80, b"[" * 80 + b"1," + b"]" * 80):
        assert _single(ctx, d) == "ok", d[:20]
    # the reference's depth error at its default limit, and at a limit below the register levels
    tape, strings, err, st = ctx.parse_document(b"[" * 1024 + b"1" + b"]" * 1024)
    assert err == 28
    tape, strings, err, st = ctx.parse_document(b"[" * 64 + b"1" + b"]" * 64, max_depth=20)
    assert err == 28
    # handed back: a depth limit above the 1024 levels the device keeps
    tape, strings, err, st = ctx.parse_document(b"[" * 1030 + b"1" + b"]" * 1030, max_depth=4096)
    assert err == NEEDS_HOST and tape is None


def test_large_array_count_saturates(ctx):
    """ArrayParsingTest.java:74-95: 0xFFFFFF + 1 elements -> count field 0xFFFFFF (33.5 M structurals, 524,288 steps)."""
    n = 0xFFFFFF + 1
    doc = b"[" + b"0," * (n - 1) + b"0]"
    tape, strings, err, st = ctx.parse_document(doc)
    assert err == 0 and (int(tape[1]) >> 32) & 0xFFFFFF == 0xFFFFFF and tape.size == 2 * n + 4
    assert np.array_equal(tape, O.parse(doc).tape)


def test_large_documents_chunk_parallel(ctx):
    """One document of more than 4096 structurals is walked by many waves (512-structural chunks: k_chunk_summary ->
    k_chunk_scan -> k_coop_walk<true> -> k_chunk_finish).  Valid documents whose containers, empty pairs, keys and
    separators straddle the chunk boundaries at every phase; the reference files cut, spliced and damaged at random
    positions (first error by position over the chunks); depth swings beyond what a chunk's export holds (the flagged
    fall-back to the single-wave sweep)."""
    rng = random.Random(977)
    docs = []
    for shift in range(0, 9):
        pre = "0," * shift
        docs.append("[" + pre + "[],{}," * 3000 + "1]")
        docs.append("[" + pre + '{"a":[],"b":{}},' * 1500 + "[]]")
        docs.append("{" + ",".join('"k%d":{"x":[%s]}' % (i, pre + "1") for i in range(1500)) + "}")
        docs.append("[" + pre + ",".join("[" * (i % 7) + str(i) + "]" * (i % 7) for i in range(3000)) + "]")
    # a staircase: the depth climbs over several chunks and comes back (exports of many levels, commas_low at every level)
    docs.append("[" + ("1," * 200 + "[") * 40 + "2" + ("]" + ",3" * 200) * 40 + "]")
    docs.append("{" + ('"a":1,' * 150 + '"n":{') * 50 + '"z":0' + ("}" + ',"b":2' * 150) * 50 + "}")
    # swings of more than 31 levels inside one chunk, and a depth of more than 63 levels across chunks
    docs.append("[" + "1," * 5000 + "[" * 40 + "7" + "]" * 40 + ",1" * 5000 + "]")
    docs.append("[" + ("1," * 300 + "[") * 62 + "2" + "]" * 62 + "]")
    docs.append("[" + ("1," * 300 + "[") * 70 + "2" + "]" * 70 + "]")
    # trailing content, missing brackets, a root that closes early, far from the start
    big = "[" + "1," * 6000 + "2]"
    docs += [big + "1", big + "]", big[:-1], big[:-1] + "}", "[" + "1," * 3000 + "2]" + ",1" * 3000, big.replace("1,", "1 ", 1),
             "[" + "1," * 5000 + "tru," + "1," * 100 + "]", "[" + "1," * 5000 + '"\\q",' + "1," * 100 + "0]",
             '{"a":' + big + ',"b"' + "}", '{"a":' + big + ',"b":}', '{"a":' + big + ',}', "7" + " 1" * 5000, '"s"' + ",1" * 5000]
    handed = 0
    for d in docs:
        handed += _single(ctx, d.encode()) == "host"
    assert handed == 0  # (the 70-level staircase too: the single-wave sweep keeps levels 64 .. 1023 in global memory)
    for name in ("twitter.json", "github_events.json"):
        base = load_fixture(name)
        for _ in range(60):
            r = rng.random()
            p = rng.randrange(len(base))
            if r < 0.25:
                d = base[:p]
            elif r < 0.5:
                d = base[:p] + rng.choice([b"]", b"}", b",", b":", b"[", b"{", b'"', b"x", b"1"]) + base[p:]
            elif r < 0.75:
                d = base[:p] + base[p + 1:]
            else:
                q = rng.randrange(len(base))
                d = base[:min(p, q)] + base[max(p, q):]
            _single(ctx, d)


def test_chunk_and_group_size_boundaries(ctx):
    """Structural counts around every switch of the chunk-parallel path: 1,024 (single-wave sweep below), multiples of the
    128- and 512-structural chunks, 262,144 (chunk length 128 -> 512), squares of the group size; and documents whose BOUND
    asks for chunks (more than 4 KiB) while they hold a handful of structurals (the summary pass hands them back)."""
    def with_structurals(s):
        # "[0,0,...,0]" has 2k + 3 structurals; one "[]," in front adds 3
        even = s % 2 == 0
        k = (s - 3 - (3 if even else 0)) // 2
        doc = "[" + ("[]," if even else "") + "0," * k + "0]"
        return doc.encode()
    counts = [1021, 1023, 1024, 1025, 1027, 1151, 1152, 1153, 1279, 1280, 1281, 2047, 2048, 2049, 8191, 8192, 8193, 8320, 8321,
              128 * 64 - 1, 128 * 64, 128 * 64 + 1, 128 * 256 + 1, 262143, 262144, 262145, 262147, 262144 + 511, 262144 + 512, 262144 + 513,
              512 * 1024 + 1, 512 * 4096 + 5]
    for s in counts:
        d = with_structurals(s)
        idx, st = ctx.stage1(d)
        assert idx.size == s, (s, idx.size)
        assert _single(ctx, d) == "ok", s
        assert _single(ctx, d[:-1]) == "ok", s           # unclosed
        assert _single(ctx, d + b" 7") == "ok", s         # trailing content in the last chunk
    for filler in (5000, 70000, 300000):
        for doc in ('["%s"]' % ("x" * filler), '{"k":"%s","n":[1,2,{"a":null}]}' % ("y" * filler), '"%s"' % ("z" * filler), " " * filler + "[1, 2]"):
            assert _single(ctx, doc.encode()) == "ok"


def test_integer_literals_of_every_length(ctx):
    """cw_primitive's branch-free integer paths: 1 .. 15 digits out of the 16-byte window, 16 .. 18 digits with the second
    window, 19 and more through the scanner -- every length, both signs, every terminator, leading zeros, a bad byte behind
    the digits, the long range's edges; inside arrays and objects so that the literal starts at every byte phase."""
    rng = random.Random(977)
    lits = []
    for nd in range(1, 23):
        for _ in range(6):
            digits = str(rng.randrange(1, 10)) + "".join(str(rng.randrange(10)) for _ in range(nd - 1))
            lits += [digits, "-" + digits]
        lits += ["9" * nd, "-" + "9" * nd, "1" + "0" * (nd - 1), "0" + "1" * (nd - 1) if nd > 1 else "0", "-0" + "7" * (nd - 1)]
    lits += ["9223372036854775807", "9223372036854775808", "-9223372036854775808", "-9223372036854775809", "999999999999999999",
             "-999999999999999999", "1000000000000000000", "123456789012345678", "1234567890123456", "12345678901234567",
             "-123456789012345", "-1234567890123456", "-12345678901234567", "-123456789012345678", "0", "-0", "-", "--1"]
    docs = []
    for lit in lits:
        for term in (",", "]", " ]", "\n]", "\t,1]"):
            docs.append(("[" + lit + term + ("" if "]" in term else "2]")).encode())
        docs.append(('{"k":' + lit + "}").encode())
        docs.append(('{"kk": ' + lit + ' ,"x":' + lit + "}").encode())
        docs.append(("[" + lit + "x]").encode())        # a byte behind the digits that is neither separator nor digit
        docs.append(("[" + lit + ".5]").encode())
        docs.append(("[" + lit + "e2]").encode())
    for d in docs[::7]:
        assert _single(ctx, d) == "ok", d
    tapes, strings, errors = gpu_walk(ctx, docs)
    check_against_oracle(docs, tapes, strings, errors)  # (no hand-backs: host_ok is empty)
