"""CPU fuzz of the GPU walker's per-document automaton (simdjson-java_amd/csrc/walk_doc.h, the code k_doc_walk runs)
against the oracle: the header is shared verbatim between walk.hip and this host build (tests/host_sim/walkdev_sim.cpp),
so grammar, atoms, integers, the exact-range float conversion and the hand-back rules are checked here without a GPU;
tests/test_gpu_walk.py then covers the kernel around it."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT, load_fixture
from tests.test_host_walk import GRAMMAR
from tests.walk_common import NEEDS_HOST, exact_range, number_documents

SIM_DIR = os.path.join(ROOT, "tests", "host_sim")


@pytest.fixture(scope="module")
def walk():
    so = os.path.join(SIM_DIR, "libwalkdevsim.so")
    deps = [os.path.join(SIM_DIR, "walkdev_sim.cpp"), os.path.join(ROOT, "simdjson-java_amd", "csrc", "walk_doc.h"),
            os.path.join(ROOT, "include", "sjmi.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, deps[0]])
    lib = C.CDLL(so)
    lib.sim_walk_device.restype = C.c_int
    lib.sim_walk_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                    C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]

    def run(doc, max_depth=1024, lead=b""):
        """The document placed behind `lead` bytes of other documents (positions, index ranges and string offsets are
        then batch-wide, as in the kernel).  -> (tape or None, code), or None if stage 1 / a string rejects it first."""
        idx, st = O.stage1(doc)
        if st:
            return None
        padded = np.frombuffer(lead + bytes(doc) + b"\n" + b"\0" * 64, dtype=np.uint8)
        sb, _, feo, _ = O.unescape_all(np.frombuffer(bytes(doc) + b"\0" * 64, dtype=np.uint8), idx)
        if feo >= 0:
            return None
        pre = 7  # structurals of "earlier documents" in front of this one's
        ix = np.concatenate([np.arange(pre, dtype=np.uint32), idx + len(lead), [0]]).astype(np.uint32)
        sbuf = np.frombuffer(b"#" * 5 + sb + b"\0" * 8, dtype=np.uint8)
        tape = np.zeros(2 * idx.size + 16, dtype=np.uint64)
        n = C.c_uint32(0)
        rc = lib.sim_walk_device(padded.ctypes.data, len(lead), len(lead) + len(doc) + 1, ix.ctypes.data, pre, pre + idx.size, ix.size,
                                 sbuf.ctypes.data, 5, 1000, max_depth, tape.ctypes.data, C.addressof(n))
        return (tape[:n.value].copy() if rc == 0 else None), rc
    return run


def _rebased(want_tape, shift):
    """the oracle's tape with its STRING payloads moved by `shift` (the walker adds the batch-wide string offset)"""
    t = want_tape.copy()
    i = 0
    while i < t.size:
        ty = int(t[i]) >> 56
        if ty == ord('"'):
            t[i] = np.uint64(int(t[i]) + shift)
        i += 2 if ty in (ord("l"), ord("d")) else 1
    return t


def _check(walk, doc, max_depth=1024, lead=b"", host_ok=None):
    got = walk(doc, max_depth, lead)
    if got is None:
        return False
    tape, rc = got
    want = O.parse(doc + b"\n", max_depth=max_depth)
    if rc == NEEDS_HOST:
        assert host_ok is not False and want.error == 0, (doc[:80], want.error)
        return True
    assert host_ok is not True, doc[:80]
    assert rc == want.error, (doc[:80], rc, want.error)
    if rc == 0:
        assert np.array_equal(tape, _rebased(want.tape, 1005)), doc[:80]
    return True


EASY = [d for d in GRAMMAR if not any(x in d for x in (b"1e23", b"1e400"))]


@pytest.mark.parametrize("doc", EASY, ids=[d[:24].decode("latin1") for d in EASY])
def test_grammar_numbers_atoms(walk, doc):
    assert _check(walk, doc, host_ok=False)
    assert _check(walk, doc, lead=b'{"x": 1}\n' * 3, host_ok=False)


@pytest.mark.parametrize("name", ["twitter.json", "github_events.json", "wide_bench.json"])
def test_reference_files(walk, name):
    assert _check(walk, load_fixture(name).rstrip(), lead=b"[1]\n")


def test_hand_back_and_depth(walk):
    # handed back: more than 19 significant digits within 10^-19 of a rounding boundary (the rest of the reference's slow
    # path is decided by the two 19-digit neighbours), nesting beyond the device stack
    from tests.walk_common import AMBIGUOUS
    for doc in [("[%s]" % a).encode() for a in AMBIGUOUS] + [("-" + AMBIGUOUS[0]).encode(), b"[" * 65 + b"]" * 65, b"[" * 64 + b"1" + b"]" * 64]:
        assert _check(walk, doc, host_ok=True), doc[:60]
    for doc in (b"[12345678901234567891e0]", b"[1.2345678901234567890123]", b"3.141592653589793238462643383279",
                b"[0.10000000000000000000000000000000000001]", b"[123456789012345678901234567890.5e-400]",
                b"[0.000000000000000000000000000000123456789012345678901234567890e+330]", b"[10000000000000000000000000000000000000001]".replace(b"]", b".0]")):
        assert _check(walk, doc, host_ok=False), doc[:60]
    # converted on the device: everything Eisel-Lemire covers -- ties, subnormals, saturation, 19-digit significands
    for doc in (b"[1e23]", b"[1e-23]", b"[0.1e400]", b"[9007199254740993.0]", b"[1.7976931348623157e308]", b"[4.9e-324]", b"[2.4e-324]",
                b"[2.2250738585072013e-308]", b"[1.7976931348623159e308]", b"[-1e999]", b"[1e-999]", b"[922337203685477580.5]",
                b"[1234567890123456789e-30]", b"[1e22]", b"[" * 64 + b"]" * 64, b"[" * 63 + b"1" + b"]" * 63, b"[9007199254740992.0]", b"[100000000000000000000.0]",
                b"[1.0000000000000000000]", b"[0.000000000000000000001]"):
        assert _check(walk, doc, host_ok=False)
    for depth in (3, 4, 5, 10):
        for doc in (b"[[[[1]]]]", b'{"a":{"b":{"c":1}}}', b"[[[[]]]]", b"[" * 10 + b"]" * 10):
            assert _check(walk, doc, max_depth=depth, host_ok=False)


def test_number_fuzz(walk):
    rng = random.Random(93)
    docs, hard, either = number_documents(rng, 30000)
    n_host = 0
    for k, d in enumerate(docs):
        assert _check(walk, d, host_ok=None if k in either else (k in hard))
        n_host += k in hard
    assert 100 < n_host < 5000


def test_fuzz_documents(walk):
    rng = random.Random(32)

    def value(d):
        r = rng.random()
        if d > 4 or r < 0.45:
            return rng.choice(['"s"', '"a\\nb"', '"é€"', '"\\u00e9"', "1", "-2.5e3", "true", "false", "null", '""', "12345678",
                               "0.000001", "1e-7", "tru", "01", "1.", "", "falsey", "nul"])
        if r < 0.7:
            return "[" + rng.choice([",", ", ", " ,"]).join(value(d + 1) for _ in range(rng.randint(0, 5))) + "]"
        return "{" + ",".join('"k%d"%s%s' % (i, rng.choice([":", ":", " : ", ""]), value(d + 1)) for i in range(rng.randint(0, 5))) + "}"
    checked = 0
    for it in range(4000):
        checked += _check(walk, value(0).encode(), lead=b"x" * (it % 37), host_ok=False)
    assert checked > 3000


def test_large_array_size_saturates(walk):
    """ArrayParsingTest.java:74-95: 0xFFFFFF + 1 elements -> count field 0xFFFFFF, through the GPU walker's automaton."""
    n = 0xFFFFFF + 1
    tape, rc = walk(b"[" + b"0," * (n - 1) + b"0]")
    assert rc == 0 and (int(tape[1]) >> 32) & 0xFFFFFF == 0xFFFFFF and tape.size == 2 * n + 4


def test_reference_number_vectors(walk):
    """NumberParsingTest.java's literal vectors through the GPU walker's automaton: the asserted value / message for every
    literal of at most 19 significant digits (ties, subnormals, saturation included: Eisel-Lemire on the device) and for the
    longer ones whose two 19-digit neighbours round alike; the others are handed back -- never a different value."""
    from tests.conftest import number_vectors
    converted = handed_back = 0
    for v in number_vectors():
        doc = v["input"].encode("utf-8")[:v.get("length")]
        got = walk(doc)
        if got is None:
            continue
        if got[1] == NEEDS_HOST:
            assert "message" not in v, v["input"][:40]
            lit = v["input"].strip().strip("[]").strip()
            assert not exact_range(lit), v["input"][:60]
            handed_back += 1
        else:
            assert _check(walk, doc), v["input"][:40]
            converted += 1
    assert converted >= 150 and 1 <= handed_back <= 4  # (two literals sit exactly on a midpoint with a tail behind the 19th digit)
