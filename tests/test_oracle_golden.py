"""Pins the CPU oracle against the reference's own literal test vectors (tests/golden/vectors.py,
transcribed from /root/reference/src/test/java/org/simdjson/*Test.java) and cross-checks its
two independent restatements of each stage-1 function against each other under seeded fuzz.
CPU only (no GPU)."""
import random
import zlib

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import load_fixture
from tests.golden import vectors as V


def _msg_for_status(st):
    # SimdJsonParser.stage1 order: UTF-8 first, then unclosed string, then unescaped chars
    if st & O.ST_UTF8:
        return V.UTF8_ERROR
    if st & O.ST_UNCLOSED:
        return V.MSG_UNCLOSED
    if st & O.ST_UNESCAPED:
        return V.MSG_UNESCAPED
    return None


@pytest.mark.parametrize("case", V.STRUCTURAL_INDEXER, ids=[c[0] for c in V.STRUCTURAL_INDEXER])
def test_structural_indexer_vectors(case):
    name, data, want_idx, want_msg, cite = case
    for fn in (lambda d: O.index_blocks(d)[:2], O.index_bytewise):
        idx, st = fn(data)
        if want_msg is not None:
            assert _msg_for_status(st) == want_msg, cite
        else:
            assert st == 0, cite
            assert idx.tolist() == want_idx, cite


def _rand_valid_utf8(rng, min_chars, max_chars):
    """Stand-in for RandomStringUtils.random(n).getBytes(UTF_8) (testutils/Utf8TestData.java:56-60)."""
    n = rng.randint(min_chars, max_chars)
    out = []
    for _ in range(n):
        r = rng.random()
        if r < 0.4:
            cp = rng.randint(0, 0x7F)
        elif r < 0.6:
            cp = rng.randint(0x80, 0x7FF)
        elif r < 0.85:
            cp = rng.randint(0x800, 0xFFFF)
            if 0xD800 <= cp <= 0xDFFF:
                cp = ord("?")
        else:
            cp = rng.randint(0x10000, 0x10FFFF)
        out.append(chr(cp))
    return "".join(out).encode("utf-8")


def _validators(data):
    return (O.utf8_lookup(data, species=64), O.utf8_lookup(data, species=32), O.utf8_strict(data))


def test_utf8_valid_random():
    rng = random.Random(20250824)
    for _ in range(300):
        d = _rand_valid_utf8(rng, 1, 1000)
        assert _validators(d) == (True, True, True)


@pytest.mark.parametrize("case", V.UTF8_INVALID_MID, ids=[c[0] for c in V.UTF8_INVALID_MID])
def test_utf8_invalid_mid(case):
    name, seq, cite = case
    rng = random.Random(zlib.crc32(name.encode()))
    for _ in range(40):
        d = _rand_valid_utf8(rng, 0, 500) + seq + _rand_valid_utf8(rng, 0, 500)
        assert _validators(d) == (False, False, False), cite


@pytest.mark.parametrize("case", V.UTF8_INVALID_END, ids=[c[0] for c in V.UTF8_INVALID_END])
def test_utf8_invalid_end(case):
    name, seq, cite = case
    rng = random.Random(zlib.crc32(name.encode()))
    for _ in range(40):
        d = _rand_valid_utf8(rng, 0, 1000) + seq
        assert _validators(d) == (False, False, False), cite


@pytest.mark.parametrize("fam", V.UTF8_INVALID_FAMILIES, ids=[f[0] for f in V.UTF8_INVALID_FAMILIES])
def test_utf8_invalid_families(fam):
    name, seqs, cite = fam
    rng = random.Random(7)
    step = max(1, len(seqs) // 1500)  # overlongFourByteSequence has 65536 members: sample + both ends
    pick = list(range(0, len(seqs), step)) + [len(seqs) - 1]
    for i in pick:
        d = _rand_valid_utf8(rng, 0, 80) + seqs[i] + _rand_valid_utf8(rng, 0, 80)
        assert _validators(d) == (False, False, False), (cite, seqs[i].hex())


def test_utf8_lookup_equals_strict_fuzz():
    """Utf8Validator's lookup algorithm == strict RFC 3629 validity, for both species widths."""
    rng = random.Random(1234)
    interesting = [0x00, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC1, 0xC2, 0xDF, 0xE0, 0xE1, 0xEC,
                   0xED, 0xEE, 0xEF, 0xF0, 0xF1, 0xF3, 0xF4, 0xF5, 0xF7, 0xF8, 0xFF, 0x41]
    for it in range(20000):
        n = rng.randint(0, 140)
        if it % 2:
            d = bytes(rng.choice(interesting) for _ in range(n))
        else:
            base = bytearray(_rand_valid_utf8(rng, 0, 60))
            for _ in range(rng.randint(0, 2)):
                if base:
                    base[rng.randrange(len(base))] = rng.choice(interesting)
            d = bytes(base)
        a, b, c = _validators(d)
        assert a == b == c, d.hex()


def test_index_block_form_equals_bytewise_fuzz():
    """index512 block algebra == per-byte state machine (alignment invariance, SURVEY 8(a) a3')."""
    rng = random.Random(99)
    alphabet = b'\\\\\\"""{}[]:, \t\n\r\x0c\x1a\x01abc019.-e\xc3\xa9'
    for it in range(6000):
        n = rng.choice([0, 1, 2, 63, 64, 65, 127, 128, 129, 200, rng.randint(0, 400)])
        mode = it % 4
        if mode == 0:
            d = bytes(rng.choice(alphabet) for _ in range(n))
        elif mode == 1:
            d = bytes(rng.choice(b'\\"a') for _ in range(n))
        elif mode == 2:  # long backslash runs across block boundaries
            d = b"a" * rng.randint(0, 70) + b"\\" * rng.randint(50, 200) + b'"' + b"x" * rng.randint(0, 70)
        else:
            d = bytes(rng.getrandbits(8) for _ in range(n))
        i1, s1, _ = O.index_blocks(d)
        i2, s2 = O.index_bytewise(d)
        assert s1 == s2 and np.array_equal(i1, i2), d.hex()
        # alignment invariance: k leading spaces shift every index by k
        k = rng.randint(1, 70)
        i3, s3, _ = O.index_blocks(b" " * k + d)
        assert s3 == s1 and np.array_equal(i3, i1 + k)


def test_len_shorter_than_buffer_is_invisible():
    """ArrayParsingTest.java:214-245 / StringParsingTest.java:262-274: bytes >= len are invisible."""
    d = b'[1,2,3]"\\{{{{' * 10
    for n in (0, 1, 7, 8, 9, 64, 65, 100):
        a = O.stage1(d, n)
        b = O.stage1(d[:n])
        assert a[1] == b[1] and np.array_equal(a[0], b[0])


@pytest.mark.parametrize("name", list(V.FILES))
def test_reference_files(name):
    size, S, st, utf8_ok, first, last = V.FILES[name]
    d = load_fixture(name)
    assert len(d) == size
    idx, status = O.stage1(d)
    assert len(idx) == S and status == st
    assert _validators(d) == (utf8_ok,) * 3
    if first:
        assert idx[:len(first)].tolist() == first and idx[-len(last):].tolist() == last
    i2, s2 = O.index_bytewise(d)
    assert np.array_equal(idx, i2) and (status & ~O.ST_UTF8) == s2


def test_malformed_file_is_invalid_utf8():
    d = load_fixture("malformed.txt")  # Utf8ValidationTest.java:436-448
    assert _validators(d) == (False, False, False)
    assert O.parse(d).message == V.UTF8_ERROR
    cut = V.MALFORMED_FIRST_BAD_OFFSET
    assert _validators(d[:cut]) == (True, True, True) and not O.utf8_strict(d[:cut + 1])


# ----------------------------------------------------------------------------------------------
# strings + stage 2
# ----------------------------------------------------------------------------------------------

def _parse_text(text, length=None):
    b = text.encode("utf-8") if isinstance(text, str) else text
    return O.parse(b, length)


@pytest.mark.parametrize("case", V.STRING_ERRORS, ids=[repr(c[0]) for c in V.STRING_ERRORS])
def test_string_error_vectors(case):
    text, msg, cite = case
    p = _parse_text(text)
    assert p.error != 0
    if msg == V.MSG_ESCAPE:
        assert p.message.startswith(msg), cite  # hasMessageStartingWith
    else:
        assert p.message == msg, cite


def test_string_misc_vectors():
    text, want, cite = V.LONG_STRING
    assert _parse_text(text).to_python() == ("a", 1, [("s", want.encode())]), cite
    text, want, cite = V.ARRAY_OF_STRINGS
    assert _parse_text(text).to_python() == ("a", 2, [("s", w.encode()) for w in want]), cite
    text, n, msg, cite = V.LEN_SHORTER
    assert _parse_text(text, n).message == msg, cite


def test_every_code_point_as_unicode_escape():
    """StringParsingTest.java:51-70 (usableEscapedUnicodeCharacters): every code point except
    surrogates, as \\uXXXX or a surrogate pair, decodes to its UTF-8 encoding."""
    for cp in list(range(0, 0x10000, 7)) + list(range(0x10000, 0x110000, 257)) + [0x7F, 0x80, 0x7FF, 0x800, 0xFFFF, 0x10000, 0x10FFFF]:
        if 0xD800 <= cp <= 0xDFFF:
            continue
        if cp < 0x10000:
            esc = "\\u%04X" % cp
        else:
            v = cp - 0x10000
            esc = "\\u%04X\\u%04X" % (0xD800 + (v >> 10), 0xDC00 + (v & 0x3FF))
        p = _parse_text('"%s"' % esc)
        assert p.error == 0, esc
        assert p.to_python() == ("s", chr(cp).encode("utf-8")), esc


def test_low_surrogate_vectors():
    for cp in range(0xDC00, 0xE000, 13):  # StringParsingTest.java:72-92
        assert _parse_text('"\\u%04X"' % cp).message == V.MSG_LOW_RESERVED
    for low in list(range(0, 0xDC00, 997)) + list(range(0xE000, 0x10000, 499)):  # :124-144
        assert _parse_text('"\\uD800\\u%04X"' % low).message == V.MSG_LOW_RANGE


def test_unescaped_control_characters():
    for c in range(0x20):  # StringParsingTest.java:207-227
        assert _parse_text(b'"' + bytes([c]) + b'"').message == V.MSG_UNESCAPED


def test_random_strings_round_trip():
    """StringParsingTest.java:22-34 with testutils/StringTestData.java:21-32 (seeded here):
    escape '"' and '\\' and code points < 0x20 (as \\uXXXX), parse, compare with the raw string."""
    rng = random.Random(4242)
    for _ in range(400):
        raw = _rand_valid_utf8(rng, 1, 200).decode("utf-8")
        esc = "".join('\\"' if ch == '"' else "\\\\" if ch == "\\" else ("\\u%04X" % ord(ch)) if ord(ch) < 0x20 else ch
                      for ch in raw)
        p = _parse_text('"%s"' % esc)
        assert p.error == 0
        assert p.to_python() == ("s", raw.encode("utf-8"))


@pytest.mark.parametrize("case", V.GRAMMAR, ids=[repr(c[0]) + str(c[1]) for c in V.GRAMMAR])
def test_grammar_vectors(case):
    text, n, msg, cite = case
    # the reference passes a buffer longer than len; SimdJsonParser.padIfNeeded then copies it
    p = _parse_text(text, n)
    if msg is None:
        assert p.error == 0, cite
    else:
        assert p.message == msg, cite


@pytest.mark.parametrize("case", V.VALID_DOCS, ids=[repr(c[0]) for c in V.VALID_DOCS])
def test_valid_doc_vectors(case):
    text, want, cite = case
    assert _parse_text(text).to_python() == want, cite


def test_twitter_default_profile_users(twitter):
    """BenchmarkCorrectnessTest.java:19-42: 86 unique screen_names with default_profile == true."""
    v = O.parse(twitter).to_python()
    statuses = dict(v[2])[b"statuses"]
    users = set()
    for tw in statuses[2]:
        u = dict(dict(tw[2])[b"user"][2])
        if u[b"default_profile"] == ("t",):
            users.add(u[b"screen_name"][1])
    assert len(users) == V.TWITTER_DEFAULT_PROFILE_USERS


def test_numbers():
    """NumberParsingTest.java -- grammar / range vectors that do not need the (absent) fxx golden files."""
    import struct

    def dbl(x):
        return ("d", struct.unpack("<Q", struct.pack("<d", x))[0])
    ok = {"0": ("l", 0), "-0": ("l", 0), "123": ("l", 123), "-9223372036854775808": ("l", -(1 << 63)),
          "9223372036854775807": ("l", (1 << 63) - 1), "1.5": dbl(1.5), "-0.0": dbl(-0.0), "1e2": dbl(100.0),
          "1E+2": dbl(100.0), "1e-2": dbl(0.01), "1e999": dbl(float("inf")), "-1e999": dbl(float("-inf")),
          "1e-999": dbl(0.0), "4.9e-324": dbl(5e-324), "1.7976931348623157e308": dbl(1.7976931348623157e308),
          "0.1": dbl(0.1), "123456789012345678901234567890.0": dbl(1.2345678901234568e29)}
    for text, want in ok.items():
        assert _parse_text(text).to_python() == want, text
        assert _parse_text("[%s]" % text).to_python() == ("a", 1, [want]), text
    bad = {"-": O.error_message(22), "-a": O.error_message(22), "01": O.error_message(23), "-01": O.error_message(23),
           "1.": O.error_message(24), "1.e5": O.error_message(24), "1e": O.error_message(25), "1e+": O.error_message(25),
           "1a": O.error_message(26), "1.5x": O.error_message(26),
           "9223372036854775808": O.error_message(27), "-9223372036854775809": O.error_message(27),
           "12345678901234567890": O.error_message(27)}
    for text, msg in bad.items():
        assert _parse_text("[%s]" % text).message == msg, text


def test_large_array_size_saturates():
    """ArrayParsingTest.java:74-95 largeArraySize: [0,0,...] with 0xFFFFFF + 1 elements reports getSize() == 0xFFFFFF
    (the 24-bit count field of the container word saturates, TapeBuilder.java:197-203)."""
    n = 0xFFFFFF + 1
    doc = b"[" + b"0," * (n - 1) + b"0]"
    assert len(doc) == n * 2 - 1 + 2
    p = O.parse(doc)
    assert p.error == 0
    assert chr(int(p.tape[1]) >> 56) == "[" and (int(p.tape[1]) >> 32) & 0xFFFFFF == 0xFFFFFF
    assert int(p.tape[1]) & 0xFFFFFFFF == p.tape.size - 1  # index behind the closing bracket


def test_reference_number_vectors():
    """Every literal vector of NumberParsingTest.java (158 inputs of 27 tests: grammar messages, long range, infinities,
    signed zeros, subnormal / normal boundaries, ties-to-even, round up / down, exponents with more digits than a long):
    the oracle's number path (strtod standing in for DoubleParser) gives the value / message the reference asserts."""
    from tests.conftest import number_vectors
    vs = number_vectors()
    assert len(vs) >= 158
    for v in vs:
        doc = v["input"].encode("utf-8")
        p = O.parse(doc, v.get("length"))
        if "message" in v:
            assert p.error != 0 and p.message == v["message"], (v["input"][:40], v["cite"], p.error)
        else:
            assert p.error == 0, (v["input"][:40], v["cite"], p.message)
            got = p.to_python()
            if "long" in v:
                assert got == ("l", v["long"]), (v["input"][:40], v["cite"], got)
            else:
                assert got == ("d", v["double_bits"]), (v["input"][:40], v["cite"], got)


# ---------------------------------------------------------------------------------------------
# the AVX-512 restatement used as bench.py's CPU timing baseline (oracle/sj_avx512.c) must give the scalar
# restatement's result -- it is only ever timed, but a baseline that computes something else would be meaningless
# ---------------------------------------------------------------------------------------------
def _avx_same(d):
    i1, s1 = O.stage1(d)
    i2, s2 = O.stage1_avx512(d)
    assert s1 == s2, (s1, s2, bytes(d[:80]).hex())
    assert np.array_equal(i1, i2)


@pytest.mark.skipif(not O.avx512_supported(), reason="host CPU without AVX-512 F+BW")
def test_avx512_restatement_equals_the_scalar_one():
    import random
    for name in ("twitter.json", "github_events.json", "wide_bench.json", "malformed.txt"):
        _avx_same(load_fixture(name))
    for case in V.STRUCTURAL_INDEXER:
        _avx_same(case[1])
    rng = random.Random(77)
    filler = ["a", "é", "€", "😀", " ", "x"]
    for _name, seq, _c in V.UTF8_INVALID_MID + V.UTF8_INVALID_END:
        for pos in (0, 1, 61, 62, 63, 64, 65, 127, 130):
            pre = "".join(rng.choice(filler) for _ in range(200)).encode()[:pos]
            while pre and (pre[-1] & 0xC0) == 0x80 or (pre and pre[-1] >= 0xC0):
                pre = pre[:-1]
            _avx_same(pre + b"x" * (pos - len(pre)) + seq)
            _avx_same(pre + b"x" * (pos - len(pre)) + seq + b"tail " * 20)
    for _name, seqs, _c in V.UTF8_INVALID_FAMILIES:
        for j in range(0, len(seqs), 37):
            _avx_same(b"y" * (j % 70) + seqs[j] + b"z" * (j % 5))
    alphabet = b'\\\\\\"""{}[]:, \t\n\r\x0c\x1a\x01abc019.-e\xc3\xa9'
    interesting = [0x00, 0x22, 0x5C, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC1, 0xC2, 0xDF, 0xE0, 0xE1,
                   0xEC, 0xED, 0xEE, 0xEF, 0xF0, 0xF1, 0xF3, 0xF4, 0xF5, 0xF7, 0xF8, 0xFF, 0x41]
    for it in range(3000):
        n = rng.choice([0, 1, 63, 64, 65, 127, 128, 129, rng.randint(0, 900)])
        mode = it % 5
        if mode == 0:
            d = bytes(rng.choice(alphabet) for _ in range(n))
        elif mode == 1:
            d = bytes(rng.choice(b'\\"a ') for _ in range(n))
        elif mode == 2:
            d = b"a" * rng.randint(0, 70) + b"\\" * rng.randint(1, 300) + rng.choice([b'"', b"x", b""]) + b'"x' * rng.randint(0, 40)
        elif mode == 3:
            d = bytes(rng.choice(interesting) for _ in range(n))
        else:
            d = bytes(rng.getrandbits(8) for _ in range(n))
        _avx_same(d)


def test_per_document_digests_see_every_kind_of_difference():
    """oracle.digest_many / digest_outputs (the all-documents check of bench.py and tests/test_gpu_fullscale.py): outputs assembled
    from the oracle's own per-document parses digest equal; one changed tape word, string byte, number bit, index or verdict does not."""
    import random
    from tests.test_gpu_batch import _small_docs
    rng = random.Random(3)
    docs = _small_docs(rng, 200) + [b"[1 1]", b'{"a":"\\u00e9\\n","b":[1.5,-2,true,null,{"c":"d"}]}']
    buf = b"".join(d + b"\n" for d in docs)
    offs = np.concatenate([[0], np.cumsum([len(d) + 1 for d in docs])]).astype(np.uint64)
    dig, err, ih, cnt = O.digest_many(buf, offs, threads=3)
    tapes, sbs, to, idxs, io, errs, sb_off = [], [], [0], [], [0], [], 0
    for k, d in enumerate(docs):
        p = O.parse(d + b"\n")
        errs.append(p.error)
        ix, _ = O.stage1(d + b"\n")
        idxs.append(ix.astype(np.uint32) + np.uint32(int(offs[k])))
        io.append(io[-1] + ix.size)
        if p.error:
            to.append(to[-1])
            continue
        tp = np.array(p.tape, dtype=np.uint64)
        i = 0
        while i < tp.size:
            t = int(tp[i]) >> 56
            if t == 0x22:
                tp[i] = np.uint64((t << 56) | ((int(tp[i]) & 0xFFFFFFFFFFFFFF) + sb_off))
            i += 2 if t in (0x6C, 0x64) else 1
        tapes.append(tp)
        sbs.append(bytes(p.strings))
        sb_off += len(p.strings)
        to.append(to[-1] + tp.size)
    tape, sb, idx = np.concatenate(tapes), np.frombuffer(b"".join(sbs), dtype=np.uint8).copy(), np.concatenate(idxs)
    to, io, errs = np.array(to, dtype=np.uint64), np.array(io, dtype=np.uint64), np.array(errs, dtype=np.int32)
    assert np.array_equal(errs, err) and errs[len(docs) - 2] != 0

    def run():
        return O.digest_outputs(tape, to, sb, idx, io, offs, errs, threads=2)
    g = run()
    assert np.array_equal(g[0], dig) and np.array_equal(g[1], ih) and np.array_equal(g[2], cnt)
    last = len(docs) - 1
    for what in ("tape word", "string byte", "index"):
        keep = (tape.copy(), sb.copy(), idx.copy())
        if what == "tape word":
            tape[int(to[last]) + 1] ^= np.uint64(1 << 33)       # the element count of the root object
        elif what == "string byte":
            sb[sb.size - 1] ^= 1
        else:
            idx[int(io[last]) + 3] += 1
        g = run()
        changed = (g[0] != dig) | (g[1] != ih)
        assert changed[last] and not changed[:last].any(), what
        tape[:], sb[:], idx[:] = keep
