"""The batch walker as a lane-level model (tools/tok_walk_model.py: the kernel's ring, steps, wave shifts, per-level words, comma
counters and stack, with the grammar table of csrc/sj_tokens.h itself) against the oracle's sequential stage 2: a document the model
keeps has the oracle's tape word for word; a document the oracle parses and whose root is a container is kept (unless it nests
deeper than the walker's 64-level stack: the exact walker's by design); every other document is handed on."""
import ctypes as C
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT, load_fixture

sys.path.insert(0, os.path.join(ROOT, "tools"))
import tok_walk_model as M  # noqa: E402
import token_docs  # noqa: E402

SIM_DIR = os.path.join(ROOT, "tests", "host_sim")


@pytest.fixture(scope="module")
def tables():
    src, lib = os.path.join(SIM_DIR, "tok_sim.cpp"), os.path.join(SIM_DIR, "libtoksim.so")
    hdr = os.path.join(ROOT, "simdjson-java_amd", "csrc", "sj_tokens.h")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", lib, src])
    L = C.CDLL(lib)
    L.sim_tok_of_first_byte.restype = L.sim_tok_grammar.restype = C.c_uint32
    L.sim_tok_of_first_byte.argtypes = L.sim_tok_grammar.argtypes = [C.c_uint32]
    first = [L.sim_tok_of_first_byte(b) for b in range(256)]
    gram = [L.sim_tok_grammar(i) for i in range(2048)]
    return (lambda b: first[b]), (lambda i: gram[i])


def run(tables, doc, max_depth=1024):
    idx, st = O.stage1(doc)
    if st:
        return None
    buf = np.frombuffer(bytes(doc) + b"\n" + b"\0" * 64, dtype=np.uint8)
    _, offs, feo, _ = O.unescape_all(np.frombuffer(bytes(doc) + b"\0" * 64, dtype=np.uint8), idx)
    if feo >= 0:
        return None
    return M.walk(tables, buf, [int(x) for x in idx], [int(o) for o in offs], 0, max_depth)


def depth_of(doc):
    d = m = 0
    for b in doc:
        if b in b"[{":
            d += 1
            m = max(m, d)
        elif b in b"]}":
            d -= 1
    return m


def check(tables, doc):
    got = run(tables, doc)
    if got is None:
        return 0
    kept, tape = got
    want = O.parse(doc + b"\n")
    if kept:
        assert want.error == 0 and [int(x) for x in want.tape] == tape, doc[:200]
        return 1
    if want.error == 0 and doc.lstrip()[:1] in (b"[", b"{"):
        assert depth_of(doc) >= 64, doc[:200]  # (only the stack's depth may send a well-formed container document on)
    return 0


def test_token_documents_against_the_oracle(tables):
    rng = random.Random(20250926)
    kept = 0
    for _ in range(1500):
        kept += check(tables, token_docs.document(rng))
    assert kept > 400


def test_step_boundaries_at_every_phase(tables):
    """the 64-token boundary in front of / behind every kind of token: a flat array with a nested tail, shifted one token at a time"""
    kept = 0
    for lead in range(55, 70):
        for tail in (b'{"a":[],"b":{},"c":[1,{"d":"e"}]}', b"[[[]],[{}]]", b'"s"', b"[]"):
            kept += check(tables, b"[" + b"1," * lead + tail + b",true]")
            kept += check(tables, b'{"k":' + b"[" * 3 + b"0," * lead + tail + b"]]]}")
    assert kept == 15 * 4 * 2


def test_twitter(tables):
    doc = load_fixture("twitter.json")
    kept, tape = run(tables, doc)
    want = O.parse(doc)
    assert kept and [int(x) for x in want.tape] == tape
