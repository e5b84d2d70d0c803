"""End-to-end GPU parity: SimdJsonParser.parse (GPU stage 1 + GPU string unescape + C++ host stage 2) against
the oracle's whole-parser restatement: raw tape words, string buffer and error messages must be identical
(BASELINE.json configs[4]: 'JsonValue equality vs reference')."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import load_fixture
from tests.golden import vectors as V

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["host_walk", "gpu_walk", "by_size"])
def parser(request):
    """The homes of stage 2: the host walker over GPU-made indexes / strings, the cooperative GPU walker
    (sjmi_parser_set_gpu_walk: the tape itself comes from the device), and the library's default -- by size, the GPU walker
    from 1 MiB on.  Every test below holds for all of them, bit for bit."""
    import simdjson_java_amd as S
    p = S.SimdJsonParser(capacity=8 * 1024 * 1024, gpu_walk={"host_walk": False, "gpu_walk": True, "by_size": None}[request.param])
    yield p
    p.close()


def _same(parser, doc, length=None):
    import simdjson_java_amd as S
    doc = doc.encode("utf-8") if isinstance(doc, str) else bytes(doc)
    want = O.parse(doc, length)
    try:
        got = parser.parse(doc, length)
    except S.JsonParsingException as e:
        assert want.error != 0, "GPU path raised %r but the oracle parsed the document" % str(e)
        assert e.code == want.error
        if want.error == 4:  # "Escaped unexpected character: " + the char (hasMessageStartingWith in the reference)
            assert str(e).startswith(want.message)
        else:
            assert str(e) == want.message
        return None
    assert want.error == 0, "oracle rejects (%s) what the GPU path parsed" % want.message
    assert np.array_equal(got.tape, want.tape)
    assert got.strings == want.strings
    return O.Parsed(got.tape, got.strings, 0, 0, 0)


@pytest.mark.parametrize("case", V.GRAMMAR, ids=[repr(c[0]) + str(c[1]) for c in V.GRAMMAR])
def test_reference_grammar_vectors(parser, case):
    import simdjson_java_amd as S
    text, n, msg, cite = case
    if msg is None:
        assert _same(parser, text, n) is not None, cite
    else:
        with pytest.raises(S.JsonParsingException) as ei:
            parser.parse(text.encode(), n)
        assert str(ei.value) == msg, cite
        _same(parser, text, n)


@pytest.mark.parametrize("case", V.STRING_ERRORS, ids=[repr(c[0]) for c in V.STRING_ERRORS])
def test_reference_string_error_vectors(parser, case):
    import simdjson_java_amd as S
    text, msg, cite = case
    with pytest.raises(S.JsonParsingException) as ei:
        parser.parse(text.encode())
    assert str(ei.value).startswith(msg), cite
    _same(parser, text)


@pytest.mark.parametrize("case", V.VALID_DOCS, ids=[repr(c[0]) for c in V.VALID_DOCS])
def test_reference_valid_docs(parser, case):
    text, want, cite = case
    got = _same(parser, text)
    assert got.to_python() == want, cite


def test_stage1_error_vectors(parser):
    import simdjson_java_amd as S
    for name, data, want_idx, want_msg, cite in V.STRUCTURAL_INDEXER:
        if want_msg:
            with pytest.raises(S.JsonParsingException) as ei:
                parser.parse(data)
            assert str(ei.value) == want_msg, cite
    for c in range(0x20):  # StringParsingTest.java:207-227
        with pytest.raises(S.JsonParsingException) as ei:
            parser.parse(b'"' + bytes([c]) + b'"')
        assert str(ei.value) == V.MSG_UNESCAPED
    with pytest.raises(S.JsonParsingException) as ei:
        parser.parse(load_fixture("malformed.txt"))
    assert str(ei.value) == V.UTF8_ERROR


@pytest.mark.parametrize("name", ["twitter.json", "github_events.json", "wide_bench.json"])
def test_reference_files(parser, name):
    doc = load_fixture(name)
    for _ in range(2):  # parser reuse (BenchmarkCorrectnessTest.java:24 parses 10x with one parser)
        got = _same(parser, doc)
    if name == "twitter.json":  # BenchmarkCorrectnessTest.java:19-42
        v = got.to_python()
        users = set()
        for tw in dict(v[2])[b"statuses"][2]:
            u = dict(dict(tw[2])[b"user"][2])
            if u[b"default_profile"] == ("t",):
                users.add(u[b"screen_name"][1])
        assert len(users) == V.TWITTER_DEFAULT_PROFILE_USERS


def test_numbers_and_nesting(parser):
    docs = ["0", "-0", "123", "-9223372036854775808", "9223372036854775807", "1.5", "-0.0", "1e2", "1E+2", "1e-2", "1e999",
            "-1e999", "1e-999", "4.9e-324", "0.1", "123456789012345678901234567890.0", "[1e5,2.5,-3]",
            "-", "01", "1.", "1e", "1a", "9223372036854775808", "-9223372036854775809", "[-]", "[01]", "[1.]", "[1e+]", "[1x]",
            "[" * 30 + "]" * 30, '{"a":' * 20 + "1" + "}" * 20, "[[],{},[{}],{\"a\":[]}]", " [ 1 , 2 ] ", "[1,2,3]   \n"]
    for d in docs:
        _same(parser, d)


def test_depth_limit():
    import simdjson_java_amd as S
    p = S.SimdJsonParser(capacity=1 << 20, max_depth=8)
    try:
        ok = "[" * 7 + "]" * 7
        assert p.parse(ok.encode()).tape.size > 0
        with pytest.raises(S.JsonParsingException) as ei:
            p.parse(("[" * 8 + "1" + "]" * 8).encode())
        assert ei.value.code == 28 and O.parse(("[" * 8 + "1" + "]" * 8).encode(), max_depth=8).error == 28
    finally:
        p.close()


def test_random_documents(parser):
    rng = random.Random(2025)

    def value(d):
        r = rng.random()
        if d > 4 or r < 0.35:
            return rng.choice(['"s"', '"a\\nb"', '"é€"', '"\\u00e9\\ud83d\\ude00"', "1", "-2.5e3", "true", "false", "null", '""',
                               '"' + "x" * rng.randint(0, 90) + '"'])
        if r < 0.65:
            return "[" + ",".join(value(d + 1) for _ in range(rng.randint(0, 6))) + "]"
        return "{" + ",".join('"k%d":%s' % (i, value(d + 1)) for i in range(rng.randint(0, 6))) + "}"
    for it in range(300):
        doc = value(0)
        if it % 5 == 0:  # corrupt one byte: any verdict must match the oracle's
            b = bytearray(doc.encode())
            if b:
                b[rng.randrange(len(b))] = rng.choice(b'{}[]:,"\\ x0')
            _same(parser, bytes(b))
        else:
            _same(parser, doc)


def test_large_array_size_saturates():
    """ArrayParsingTest.java:74-95 largeArraySize: [0,0,...] with 0xFFFFFF + 1 elements (33.5 MB, 33.5 M structurals) parses
    and reports getSize() == 0xFFFFFF; tape equal to the oracle's."""
    import simdjson_java_amd as S
    n = 0xFFFFFF + 1
    doc = b"[" + b"0," * (n - 1) + b"0]"
    p = S.SimdJsonParser(capacity=len(doc) + 64)
    try:
        got = p.parse(doc)
        want = O.parse(doc)
        assert want.error == 0 and (int(got.tape[1]) >> 32) & 0xFFFFFF == 0xFFFFFF
        assert np.array_equal(got.tape, want.tape)
    finally:
        p.close()


def test_reference_number_vectors(parser):
    """NumberParsingTest.java's 158 literal vectors end to end on the GPU path: the value / message the reference asserts."""
    import simdjson_java_amd as S
    from tests.conftest import number_vectors
    for v in number_vectors():
        doc = v["input"].encode("utf-8")
        if "message" in v:
            with pytest.raises(S.JsonParsingException) as e:
                parser.parse(doc, v.get("length"))
            assert str(e.value) == v["message"], (v["input"][:40], v["cite"])
        else:
            got = O.Parsed(parser.parse(doc, v.get("length")).tape, b"", 0, 0, 0).to_python()
            assert got == (("l", v["long"]) if "long" in v else ("d", v["double_bits"])), (v["input"][:40], v["cite"], got)


def test_json_value_accessors_through_the_c_abi(parser, twitter):
    """The JsonValue the boundary advertises (sjmi_value_* -> C++ org_simdjson::JsonValue, JsonValue.java:25-111):
    BenchmarkCorrectnessTest.java:19-42 written against it -- get("statuses").arrayIterator(), get("user"),
    asBoolean(), asString() -> 86 users -- and whole trees rebuilt through isX / asX / getSize / iterators equal to
    the oracle's trees (type, int64, raw double bits, UTF-8 bytes, order, size)."""
    import simdjson_java_amd as S
    for _ in range(2):
        parser.parse(twitter)
        root = parser.root()
        assert root.isObject() and not root.isArray() and root.get("no such field") is None
        users = set()
        n_tweets = 0
        for tweet in root.get("statuses").arrayIterator():
            n_tweets += 1
            user = tweet.get("user")
            if user.get("default_profile").asBoolean():
                users.add(user.get("screen_name").asString())
        assert len(users) == V.TWITTER_DEFAULT_PROFILE_USERS and n_tweets == root.get("statuses").getSize() == 100
        assert root.get("search_metadata").get("count").asLong() == 100
        with pytest.raises(S.SjmiError):
            root.get("statuses").asLong()  # wrong type: Java would throw
    for name in ("twitter.json", "github_events.json", "wide_bench.json"):
        doc = load_fixture(name)
        parser.parse(doc)
        assert parser.root().to_python() == O.parse(doc).to_python(), name
    for text in ["[]", "{}", "[[]]", '{"a":{}}', "[1,-2,3.5,1e300,true,false,null,\"x\\u00e9\\ud83d\\ude00\",\"\"]", "7", '"s"', "null",
                 '{"k":[{"a":1},{"b":[2,3]}],"é":"€"}']:
        parser.parse(text.encode())
        assert parser.root().to_python() == O.parse(text.encode()).to_python(), text


def test_documents_around_the_stage2_placement_threshold(parser):
    """128 KiB is where the default placement of stage 2 changes (SimdJsonParser::GPU_WALK_AUTO_BYTES; 1 MiB in round 2): documents
    just below and above it and larger ones, valid and broken (a broken one is walked again on the host for the reference's exact
    message)."""
    unit = '{"id":%d,"name":"user \\u00e9 %d","tags":["a","b"],"score":%d.5,"ok":true}'
    for n in (1800, 2100, 13000, 15500, 40000):
        body = ",".join(unit % (i, i, i % 97) for i in range(n))
        _same(parser, "[" + body + "]")
        _same(parser, "[" + body + ",]")
        _same(parser, "[" + body + "] 1")
        _same(parser, "[" + body.replace('"ok":true', '"ok":tru', 1) + "]")
        _same(parser, "[" + body[: len(body) // 2])


@pytest.mark.parametrize("mode", [False, True])
def test_large_documents_through_the_staged_upload(mode):
    """sjmi_set_input_staging (the parser's padded buffer): a document of 4 MiB .. 16 MiB goes up in 2 MiB chunks whose PCIe
    transfers overlap the copy of the next chunk, a larger one with a second copying thread -- the tape must be the oracle's,
    also when the caller's buffer changes between calls (nothing may be read from it after the call returned)."""
    import simdjson_java_amd as S
    rng = random.Random(5)

    def array(n):
        return b"[" + b",".join(b'{"id":%d,"name":"user %d \\u00e9","tags":["a","b\\n"],"score":%d.5,"ok":true}' % (rng.randrange(10**12), i, i % 97)
                                for i in range(n)) + b"]"
    p = S.SimdJsonParser(capacity=24 * 1024 * 1024, gpu_walk=mode)
    try:
        for n in (60000, 70000, 230000):          # ~5.2 MiB, ~6 MiB (not a multiple of the chunk), ~20 MiB
            doc = bytearray(array(n))
            want = O.parse(bytes(doc))
            got = p.parse(bytes(doc))
            assert want.error == 0 and np.array_equal(got.tape, want.tape), n
            assert got.strings == want.strings
    finally:
        p.close()
