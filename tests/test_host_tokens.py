"""CPU check of the token walker's tables (simdjson-java_amd/csrc/sj_tokens.h: what a structural's first byte makes of it, and the
token grammar that k_tok_stream reads from LDS) against the oracle's stage 2 (JsonIterator.java:68-193 restated in oracle/sj_oracle.c).

tests/host_sim/tok_sim.cpp walks a document's structurals sequentially with the kernel's rules around those tables; the header is
shared verbatim with the HIP kernel.  Checked here without a GPU, over EVERY sequence of up to six tokens (seven with
SJMI_LONG_TESTS=1: 5.4 M documents, 2.5 minutes -- it passes) from
{ [ { ] } "s" 1 true , : }: the token walker keeps a document exactly when the reference parses it and its root is a container
(any other root goes to the exact walker by design), and then it predicts the reference's tape length."""
import ctypes as C
import itertools
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT

SIM_DIR = os.path.join(ROOT, "tests", "host_sim")
TOKENS = [b"[", b"{", b"]", b"}", b'"s"', b"1", b"true", b",", b":"]


@pytest.fixture(scope="module")
def sim():
    src, lib = os.path.join(SIM_DIR, "tok_sim.cpp"), os.path.join(SIM_DIR, "libtoksim.so")
    hdr = os.path.join(ROOT, "simdjson-java_amd", "csrc", "sj_tokens.h")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", lib, src])
    L = C.CDLL(lib)
    L.sim_tok_walk.restype = C.c_int
    L.sim_tok_walk.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    return L


def walk(sim, doc, max_depth=1024, levels=64):
    idx, st = O.stage1(doc)
    assert st == 0
    padded = np.frombuffer(bytes(doc) + b"\0" * 64, dtype=np.uint8)
    pos = np.ascontiguousarray(idx.astype(np.uint32))
    words = C.c_uint32(0)
    kept = sim.sim_tok_walk(padded.ctypes.data, pos.ctypes.data, pos.size, max_depth, levels, C.byref(words))
    return kept, words.value


def test_every_short_token_sequence(sim):
    checked = kept_n = 0
    longest = 7 if os.environ.get("SJMI_LONG_TESTS") else 6
    for n in range(1, longest + 1):
        for seq in itertools.product(range(len(TOKENS)), repeat=n):
            doc = b" ".join(TOKENS[t] for t in seq)
            kept, words = walk(sim, doc)
            r = O.parse(doc)
            container_root = seq[0] in (0, 1)
            if r.error == 0 and container_root:
                assert kept == 1 and words == len(r.tape), (doc, kept, words, len(r.tape))
                kept_n += 1
            else:
                assert kept == 0, (doc, r.error)
            checked += 1
    assert checked == sum(9 ** n for n in range(1, longest + 1)) and kept_n >= 30, kept_n


def test_first_byte_table(sim):
    """numbers take two tape words, everything else one; brackets move the depth field; separators never become tokens"""
    for doc, words in ((b"[1]", 6), (b"[-1]", 6), (b"[true]", 5), (b'["a"]', 5), (b"[[],{}]", 8), (b"[null,1.5,false]", 8)):
        kept, w = walk(sim, doc)
        assert kept == 1 and w == words == len(O.parse(doc).tape), (doc, w)


def test_depth_limits(sim):
    deep = lambda d: b"[" * d + b"]" * d
    for max_depth, levels, d, want in ((1024, 64, 63, 1), (1024, 64, 64, 1), (1024, 64, 65, 0), (8, 64, 8, 1), (8, 64, 9, 0)):
        kept, _ = walk(sim, deep(d), max_depth, levels)
        # the innermost pair is an empty container (one value, no level of its own): d brackets are d - 1 levels
        assert kept == want, (max_depth, levels, d, kept)
        if want:
            assert O.parse(deep(d), max_depth=max_depth).error == 0
