"""CPU check of the streaming string pass's block algebra (simdjson-java_amd/csrc/sj_strings.h) against the oracle's
StringParser restatement (oracle/sj_oracle.c::sjo_parse_string, StringParser.java:18-68).

tests/host_sim/str_sim.cpp chains the per-block algebra sequentially; the header is shared verbatim with the HIP kernel
(csrc/strings.hip), so what is checked here without a GPU is: which bytes are kept / dropped / patched, the headers,
the \\uXXXX look-back across block boundaries, and the error of every failing string."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT, load_fixture

SIM_DIR = os.path.join(ROOT, "tests", "host_sim")


def expected_records(doc):
    """Every string literal of the document (unescaped opening quote), in order: (position, record bytes | None, code)."""
    doc = bytes(doc)
    padded = np.frombuffer(doc + b"\0" * 128, dtype=np.uint8)
    _, _, masks = O.index_blocks(doc, want_masks=True)
    recs = []
    tmp = np.zeros(len(doc) + 64, dtype=np.uint8)
    lib = O.lib()
    for b in range(masks.shape[0]):
        opens = int(masks[b, 1]) & int(masks[b, 2])  # quote & inString: the opening quotes (StructuralIndexer.java:232-234)
        while opens:
            low = opens & -opens
            pos = 64 * b + low.bit_length() - 1
            opens ^= low
            r = lib.sjo_parse_string(padded.ctypes.data, pos, tmp.ctypes.data, 0, tmp.size)
            if r < 0:
                recs.append((pos, None, int(-r)))
            else:
                recs.append((pos, tmp[:r].tobytes(), 0))
    return recs


def check_against_oracle(doc, run):
    """run(doc) -> (string buffer bytes, record offsets, first_error or None as (byte position, code), unclosed)."""
    doc = bytes(doc)
    recs = expected_records(doc)
    sb, soff, first_error, unclosed = run(doc)
    assert not unclosed
    assert len(soff) == len(recs), (len(soff), len(recs))
    failing = [k for k, r in enumerate(recs) if r[1] is None]
    for k, (pos, rec, code) in enumerate(recs):
        off = int(soff[k])
        if rec is None:
            assert sb[off:off + 4] == bytes([0xFF, 0xFF, 0xFF, code]), (k, pos, code, sb[off:off + 4].hex(), doc[max(0, pos - 20):pos + 40])
        else:
            assert sb[off:off + len(rec)] == rec, (k, pos, sb[off:off + len(rec)], rec, doc[max(0, pos - 20):pos + 40])
            if k + 1 < len(recs):
                assert int(soff[k + 1]) == off + len(rec)
            else:
                assert len(sb) == off + len(rec)
    if not failing:
        assert first_error is None
        assert sb == b"".join(r[1] for r in recs)  # the reference's stringBuffer, byte for byte
    else:
        k = failing[0]
        assert first_error is not None
        epos, ecode = first_error
        assert ecode == recs[k][2]
        hi = recs[k + 1][0] if k + 1 < len(recs) else len(doc)
        assert recs[k][0] < epos <= hi, (recs[k][0], epos, hi)
        # everything in front of the first failing string is the reference's buffer
        want = b"".join(r[1] for r in recs[:k])
        assert sb[:len(want)] == want


@pytest.fixture(scope="module")
def sim():
    so = os.path.join(SIM_DIR, "libstrsim.so")
    src = os.path.join(SIM_DIR, "str_sim.cpp")
    hdrs = [os.path.join(ROOT, "simdjson-java_amd", "csrc", h) for h in ("sj_block.h", "sj_strings.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    lib = C.CDLL(so)
    lib.sim_strings.restype = C.c_int
    lib.sim_strings.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_uint64, C.c_void_p, C.c_int]

    def make(force_u):
        def run(doc):
            a = np.frombuffer(bytes(doc) + b"\0" * 128, dtype=np.uint8)
            cap = 2 * len(doc) + 256
            sb = np.zeros(cap, dtype=np.uint8)
            soff = np.zeros(len(doc) + 2, dtype=np.uint64)
            total, ns, fe = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
            rc = lib.sim_strings(a.ctypes.data, len(doc), sb.ctypes.data, cap, C.byref(total), C.byref(ns), soff.ctypes.data,
                                 soff.size, C.byref(fe), force_u)
            assert rc in (0, 1), rc
            err = None if fe.value == 2**64 - 1 else (fe.value >> 8, fe.value & 0xFF)
            return sb[:total.value].tobytes(), soff[:ns.value].copy(), err, rc == 1
        return run
    return make(0), make(1)


ESC = ['\\"', "\\\\", "\\/", "\\b", "\\f", "\\n", "\\r", "\\t", "\\u00e9", "\\uD83D\\uDE00", "\\u0000", "\\u0041", "\\u07FF",
       "\\u0800", "\\uFFFF", "\\ud800\\udc00", "\\uDBFF\\uDFFF", "\\u12aB"]
BAD = ["\\q", "\\u12G4", "\\uD800", "\\uD800x", "\\uD800\\n", "\\uD800\\u0041", "\\uDC00", "\\uD800\\u12", "\\u", "\\u1", "\\u12",
       "\\u123", "\\uD800\\", "\\uD800\\u", "\\uD800\\uDC0", "\\\xc4\x85", "\\uD800\\uD800", "\\U0041", "\\uDFFF\\uD800"]
CHARS = ["a", "b", " ", "é", "€", "한", "😀", "x" * 7, "y" * 70, "{", "]", ":", ",", "u", "\\\\\\\\", "0"]


def _random_doc(rng, n_strings, p_esc, p_bad, sep_choices):
    parts = []
    for _ in range(n_strings):
        n = rng.choice([0, 1, 2, 3, 5, 9, 20, 63, 64, 65, 130])
        body = []
        for _ in range(n):
            r = rng.random()
            if r < p_bad:
                body.append(rng.choice(BAD))
            elif r < p_bad + p_esc:
                body.append(rng.choice(ESC))
            else:
                body.append(rng.choice(CHARS))
        parts.append('"' + "".join(body) + '"')
    doc = ""
    for p in parts:
        doc += p + rng.choice(sep_choices)
    return doc.encode("utf-8", "surrogatepass")


def test_reference_files(sim):
    for run in sim:
        for name in ("twitter.json", "github_events.json", "wide_bench.json"):
            check_against_oracle(load_fixture(name), run)


def test_every_code_point_escape(sim):
    """StringParsingTest.java:51-70: every code point as \\uXXXX / surrogate pair, at every block phase."""
    parts = []
    for cp in list(range(0, 0x10000, 7)) + list(range(0x10000, 0x110000, 1013)) + [0x7F, 0x80, 0x7FF, 0x800, 0xFFFF, 0x10FFFF]:
        if 0xD800 <= cp <= 0xDFFF:
            continue
        if cp < 0x10000:
            parts.append('"\\u%04X"' % cp if cp % 2 else '"\\u%04x"' % cp)
        else:
            v = cp - 0x10000
            parts.append('"\\u%04X\\u%04X"' % (0xD800 + (v >> 10), 0xDC00 + (v & 0x3FF)))
    for pad in range(0, 64, 5):
        check_against_oracle((" " * pad + "[" + ",".join(parts) + "]").encode(), sim[0])
    check_against_oracle(("[" + ",".join(parts) + "]").encode(), sim[1])


def test_valid_strings_fuzz(sim):
    rng = random.Random(20250925)
    for it in range(300):
        doc = _random_doc(rng, rng.randint(1, 60), rng.choice([0.0, 0.05, 0.3, 0.9]), 0.0, [",", ", ", ":", "\n", "", " " * 17])
        check_against_oracle(doc, sim[it & 1])


def test_failing_strings_fuzz(sim):
    rng = random.Random(99)
    checked = 0
    for it in range(900):
        doc = _random_doc(rng, rng.randint(1, 40), rng.choice([0.0, 0.2, 0.6]), rng.choice([0.01, 0.05, 0.3]), [",", "", "\n", " " * 5])
        if O.stage1(doc)[1] != 0:
            continue  # (a malformed escape swallowed a closing quote: not a stage-1-valid document)
        checked += 1
        check_against_oracle(doc, sim[it & 1])
    assert checked > 300


def test_every_error_at_every_block_phase(sim):
    """every malformed escape, followed by every kind of neighbour, slid across the 64-byte block boundary"""
    follow = ['"', 'a"', 'ab"', 'abc"', '\\n"', '",""', '"\n""', '","ab"', '"x"y"']
    for bad in BAD:
        for f in follow:
            for pad in list(range(40, 70)) + [124, 127, 128]:
                doc = (" " * pad + '"' + bad + f + ',"ok\\n"').encode("utf-8", "surrogatepass")
                idx, st = O.stage1(doc)
                if st != 0:
                    continue  # (an odd number of quotes: not a stage-1-valid document)
                check_against_oracle(doc, sim[0])


def test_adjacent_strings_and_backslash_runs(sim):
    for run in sim:
        check_against_oracle(b'"a""b"', run)
        check_against_oracle(b'["",""]', run)
        check_against_oracle(b'""""""""', run)
        check_against_oracle(b'"ab""cd""e""""f"', run)
        check_against_oracle(b'x"abc"', run)  # an opening quote that is no structural still makes a record
        check_against_oracle(b"", run)
        check_against_oracle(b"[1,2,3]", run)
        for n in (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 127, 129, 200):
            for pad in (0, 1, 30, 60, 61, 62, 63):
                check_against_oracle(b" " * pad + b'"' + b"\\\\" * n + b'","' + b"\\\\" * n + b'\\""', run)
