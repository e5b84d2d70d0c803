"""CPU test of the host stage 2 (DocWalker in csrc/host/simdjson_parser.cpp: the C++ mirror of JsonIterator.walkDocument
+ TapeBuilder + the number grammar) without a GPU: the engine ABI is stubbed (tests/host_sim/walk_sim.cpp), indexes and
string records come from the oracle, the resulting tape / error must equal the oracle's own stage 2 word for word."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT, load_fixture

SIM_DIR = os.path.join(ROOT, "tests", "host_sim")


@pytest.fixture(scope="module")
def walk():
    so = os.path.join(SIM_DIR, "libwalksim.so")
    deps = [os.path.join(SIM_DIR, "walk_sim.cpp"), os.path.join(ROOT, "simdjson-java_amd", "csrc", "host", "simdjson_parser.cpp"),
            os.path.join(ROOT, "simdjson-java_amd", "csrc", "host", "simdjson_parser.h"), os.path.join(ROOT, "include", "sjmi.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so, deps[0]])
    lib = C.CDLL(so)
    lib.sim_walk.restype = C.c_int
    lib.sim_walk.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
    assert lib.sim_parser_create_fails() == 1  # no engine, no parser: there is no CPU fallback in the product

    def run(doc, max_depth=1024):
        """-> (tape np.uint64 or None, error code) of the host walker, or None if the oracle rejects a string first."""
        idx, st = O.stage1(doc)
        if st:
            return None
        padded = np.frombuffer(bytes(doc) + b"\0" * 64, dtype=np.uint8)
        sb, _, feo, _ = O.unescape_all(padded, idx)
        if feo >= 0:
            return None  # (a failing string: its marker record is the GPU kernels' business, tests/test_gpu_parse.py)
        sbuf = np.frombuffer(sb + b"\0" * 8, dtype=np.uint8)
        ix = np.concatenate([idx, [0]]).astype(np.uint32)
        cap = 2 * idx.size + 16
        tape = np.zeros(cap, dtype=np.uint64)
        n = C.c_uint64(0)
        rc = lib.sim_walk(padded.ctypes.data, len(doc), ix.ctypes.data, idx.size, sbuf.ctypes.data, max_depth, tape.ctypes.data, cap,
                          C.addressof(n))
        return (tape[:n.value].copy() if rc == 0 else None), rc
    return run


def _check(walk, doc, max_depth=1024):
    got = walk(doc, max_depth)
    if got is None:
        return False
    tape, rc = got
    want = O.parse(doc, max_depth=max_depth)
    assert rc == want.error, (doc[:80], rc, want.error)
    if rc == 0:
        assert np.array_equal(tape, want.tape), doc[:80]
    return True


GRAMMAR = [b"[1 1]", b"[1,,1]", b'{"a" 1}', b"[1,2", b'{"a":1,}', b"tru", b"[01]", b"1 2", b"[-]", b"", b"{", b"}", b"[}", b"{]",
           b'{"a":}', b"{1:2}", b"[1,]", b'{"a":1 "b":2}', b"[[[]]", b"[]]", b'{"a":{}', b"nul", b"nulll", b"falsee", b"[tru]",
           b"[nul]", b"[fals]", b"[truex]", b"[+1]", b"[.5]", b"[1.]", b"[1.e3]", b"[1e]", b"[1e+]", b"[--1]", b"[1a]",
           b"[9223372036854775808]", b"[-9223372036854775809]", b"[12345678901234567890123]", b"[1] x", b"{} {}", b"true", b"false",
           b"null", b"0", b"-0", b"-0.0", b"1", b"-1", b"1.5", b"1e3", b"1E3", b"1e+3", b"1e-3", b"123.456e-2", b'"root \\n"', b'""',
           b"[]", b"{}", b"[[]]", b"[{}]", b'{"a":[]}', b'{"a":{}}', b"[9223372036854775807]", b"[-9223372036854775808]",
           b"[0.1, 0.2, 1e22, 1e-22, 1e23, 1.7976931348623157e308, 4.9e-324, 1e400, -1e400, 0.1e-400]", b" [1, 2] ",
           b'\t{"k" : [true, false, null]}\r', b"12345 ", b"true ", b"truee", b"-", b"1.", b"[" * 40 + b"]" * 40]


@pytest.mark.parametrize("doc", GRAMMAR, ids=[d[:24].decode("latin1") for d in GRAMMAR])
def test_grammar_numbers_atoms(walk, doc):
    assert _check(walk, doc)


@pytest.mark.parametrize("name", ["twitter.json", "github_events.json", "wide_bench.json"])
def test_reference_files(walk, name):
    assert _check(walk, load_fixture(name))


def test_depth_limit(walk):
    for depth in (3, 4, 5):
        for doc in (b"[[[[1]]]]", b'{"a":{"b":{"c":1}}}', b"[[[[]]]]"):
            assert _check(walk, doc, max_depth=depth)


def test_fuzz_documents(walk):
    rng = random.Random(31)

    def value(d):
        r = rng.random()
        if d > 4 or r < 0.45:
            return rng.choice(['"s"', '"a\\nb"', '"é€"', '"\\u00e9"', "1", "-2.5e3", "true", "false", "null", '""', "12345678",
                               "0.000001", "1e-7", "123456789012345678", "tru", "01", "1.", ""])
        if r < 0.7:
            return "[" + rng.choice([",", ", ", " ,"]).join(value(d + 1) for _ in range(rng.randint(0, 5))) + "]"
        return "{" + ",".join('"k%d"%s%s' % (i, rng.choice([":", ":", " : ", ""]), value(d + 1)) for i in range(rng.randint(0, 5))) + "}"
    checked = 0
    for _ in range(3000):
        checked += _check(walk, value(0).encode())
    assert checked > 2500


def test_large_array_size_saturates(walk):
    """ArrayParsingTest.java:74-95: 0xFFFFFF + 1 elements -> count field 0xFFFFFF, through the host walker."""
    n = 0xFFFFFF + 1
    tape, rc = walk(b"[" + b"0," * (n - 1) + b"0]")
    assert rc == 0 and (int(tape[1]) >> 32) & 0xFFFFFF == 0xFFFFFF and tape.size == 2 * n + 4


def test_reference_number_vectors(walk):
    """NumberParsingTest.java's literal vectors through the host walker (its doubles come from strtod): same tape / error
    as the oracle, which tests/test_oracle_golden.py pins to the values the reference asserts."""
    from tests.conftest import number_vectors
    n = 0
    for v in number_vectors():
        doc = v["input"].encode("utf-8")[:v.get("length")]
        n += _check(walk, doc)
    assert n >= 150


def test_literals_at_rounding_boundaries_are_decided_exactly(walk):
    """DoubleParser.slowlyParseDouble (DoubleParser.java:216-330): more than 19 significant digits, within 10^-19 of the
    midpoint of two doubles -- the exact comparison of csrc/sj_bigdec.h (big integers, no libc) against the oracle's strtod:
    exact midpoints (ties to even), midpoints nudged up / down far behind the 19th digit, more than 800 digits (the digits
    left out only count as "something more"), subnormals, the neighbourhood of the largest double and of zero."""
    import struct
    from decimal import Decimal, getcontext
    from tests.walk_common import AMBIGUOUS
    getcontext().prec = 2400
    rng = random.Random(2026)
    lits = list(AMBIGUOUS)
    for exp in [0, 1, 2, 52, 500, 1000, 1021, 1022, 1023, 1024, 1076, 1500, 2044, 2045, 2046]:
        for _ in range(12):
            bits = (rng.getrandbits(52) if exp else rng.randrange(0, 1 << 52)) | (exp << 52)
            if exp == 2046:
                bits = (exp << 52) | ((1 << 52) - 1)  # the largest double: its "successor" is infinity
            lo = Decimal(struct.unpack("<d", struct.pack("<Q", bits))[0])
            hi = Decimal(struct.unpack("<d", struct.pack("<Q", bits + 1))[0]) if exp != 2046 else Decimal(2) ** 1024
            text = format((lo + hi) / 2, "f")
            if "." not in text:
                text += ".0"
            lits.append(text)                                   # the tie itself
            lits.append(text + "0" * rng.choice([1, 30, 900]) + "1")  # a hair above (also: far behind the 800th digit)
            t = text.rstrip("0")
            if t[-1] != ".":
                lits.append(t[:-1] + str(int(t[-1]) - 1) + "9" * rng.choice([12, 60, 1000]))  # a hair below
            if exp > 1100:  # the same with an exponent instead of leading digits
                mant, _, frac = text.partition(".")
                lits.append(mant[0] + "." + mant[1:] + frac + "e%d" % (len(mant) - 1))
    checked = 0
    for lit in lits:
        for sign in ("", "-"):
            checked += _check(walk, ("[" + sign + lit + "]").encode())
            checked += _check(walk, (sign + lit).encode())  # as a root value (TapeBuilder.java:183-189: the padded copy)
    assert checked == 4 * len(lits) and len(lits) > 500
