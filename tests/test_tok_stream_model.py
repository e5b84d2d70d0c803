"""The STREAM form of the batch walker as a model (tools/tok_stream_model.py: a run of consecutive documents walked as one token
stream; DESIGN.md 9 "what a next round would try" -- nothing of it is built in HIP) against the oracle, document by document: runs
that mix well-formed and broken documents, documents that end or begin inside a token step, documents of one token, broken
documents directly in front of well-formed ones (the isolation of a failing document is the point of the design)."""
import ctypes as C
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import tok_stream_model as M  # noqa: E402
import token_docs  # noqa: E402
from tests.test_tok_walk_model import depth_of, tables  # noqa: E402,F401


def usable(doc):
    """passes stage 1 on its own and has no string the StringParser would throw on (the model does not model those)"""
    idx, st = O.stage1(doc)
    if st:
        return False
    _, _, feo, _ = O.unescape_all(np.frombuffer(bytes(doc) + b"\0" * 64, dtype=np.uint8), idx)
    return feo < 0


def make_run(docs):
    buf = b"\n".join(docs) + b"\n"
    padded = np.frombuffer(buf + b"\0" * 64, dtype=np.uint8)
    structurals, metas = [], []
    off = toff = ordinal = 0
    for d in docs:
        idx, st = O.stage1(d)
        assert st == 0
        frm = len(structurals)
        structurals += [off + int(x) for x in idx]
        words, strings = 2, 0
        for x in idx:
            b = d[int(x)]
            if b in b",:":
                continue
            words += 2 if (b == 0x2D or 0x30 <= b <= 0x39) else 1
        i, in_str = 0, False                     # opening quotes of the document, byte level (what the string pass counts)
        while i < len(d):
            if d[i] == 0x5C and in_str:
                i += 2
                continue
            if d[i] == 0x22:
                if not in_str:
                    strings += 1
                in_str = not in_str
            i += 1
        metas.append(dict({"from": frm, "to": len(structurals), "dso": ordinal, "toff": toff, "room": words}))
        toff += words
        ordinal += strings
        off += len(d) + 1
    return padded, structurals, metas, [16 * r + 4 for r in range(ordinal + 64)]


def check_run(tables, docs):
    padded, structurals, metas, soff = make_run(docs)
    got = M.walk_run(tables, padded, structurals, metas, soff)
    kept_n = 0
    for j, d in enumerate(docs):
        kept, tape = got[j]
        want = O.parse(d + b"\n")
        if kept:
            assert want.error == 0, (j, d[:120])
            exp, k = [int(x) for x in want.tape], 0
            for i, w in enumerate(exp):
                if w >> 56 == 0x22:
                    exp[i] = (0x22 << 56) | soff[metas[j]["dso"] + k]
                    k += 1
            assert tape == exp, (j, d[:120])
            kept_n += 1
        elif want.error == 0 and d.lstrip()[:1] in (b"[", b"{"):
            assert depth_of(d) >= 64, (j, d[:120])
    return kept_n


def test_runs_of_token_documents(tables):
    rng = random.Random(4242)
    kept = total = 0
    for _ in range(400):
        docs = []
        while len(docs) < rng.choice([1, 2, 5, 17, 32, 64]):
            d = token_docs.document(rng)
            if usable(d) and len(d) < 3000:
                docs.append(d)
        kept += check_run(tables, docs)
        total += len(docs)
    assert kept > total // 3


def test_small_and_broken_neighbours(tables):
    """one-token documents, a broken document directly in front of / behind a well-formed one, at every offset inside a step"""
    good = [b"[]", b"{}", b"[1]", b'{"a":[1,2,{"b":"c"}]}', b'["x","y"]', b"[[[[1]]]]", b'{"k":{"k":{"k":[]}}}']
    bad = [b"[", b"]", b"[1,", b"[1 2]", b'{"a"}', b'{"a":1,}', b"[1],2", b"[] []", b",", b":", b"[,]", b"1", b'"s"', b"[1]]", b"[[1]",
           b'{"k":2"v"}', b"[tru]", b"[01]", b"[1],", b",[1]"]
    rng = random.Random(7)
    for pad in range(0, 70, 3):
        filler = b"[" + b",".join([b"1"] * pad) + b"]" if pad else b"[]"
        for b in bad:
            docs = [filler, rng.choice(good), b, rng.choice(good), b, b, rng.choice(good)]
            docs = [d for d in docs if usable(d)]
            kept = check_run(tables, docs)
            assert kept >= sum(1 for d in docs if d in good or d == filler), (pad, b)


def test_a_run_of_well_formed_documents_is_kept_whole(tables):
    rng = random.Random(99)
    for _ in range(40):
        docs = []
        while len(docs) < 64:
            d = token_docs.document(rng)
            if len(d) < 2000 and usable(d) and O.parse(d + b"\n").error == 0 and d[:1] in (b"[", b"{") and depth_of(d) < 64:
                docs.append(d)
        assert check_run(tables, docs) == 64
