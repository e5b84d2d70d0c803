"""The cooperative (scan-based) stage 2 planned for the next round, as a model (tools/coop_walk_model.py): local grammar
predicates + "lowest failing position wins" + tapes from prefix sums must reproduce the oracle's sequential walker --
error code for error code, tape word for tape word."""
import os
import random
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT, load_fixture
from tests.test_host_walk import GRAMMAR

sys.path.insert(0, os.path.join(ROOT, "tools"))
import coop_walk_model as M  # noqa: E402


def _run(doc, max_depth=1024):
    idx, st = O.stage1(doc)
    if st:
        return None
    padded = bytes(doc) + b"\n" + b"\0" * 64
    a = np.frombuffer(padded, dtype=np.uint8)
    sb, offs, feo, fec = O.unescape_all(np.frombuffer(bytes(doc) + b"\0" * 64, dtype=np.uint8), idx)
    if feo >= 0:
        return None
    sizes = [0] * idx.size
    quotes = [i for i in range(idx.size) if doc[int(idx[i])] == 0x22]
    ends = list(offs[1:]) + [len(sb)]
    for q, (o, e) in zip(quotes, zip(offs, ends)):
        sizes[q] = int(e) - int(o)
    return M.walk(a, 0, len(doc) + 1, [int(x) for x in idx], sizes, [0] * idx.size, max_depth)


def _check(doc, max_depth=1024):
    got = _run(doc, max_depth)
    if got is None:
        return False
    code, tape = got
    want = O.parse(doc + b"\n", max_depth=max_depth)
    assert code == want.error, (doc[:80], code, want.error)
    if code == 0:
        assert np.array_equal(tape, want.tape), doc[:80]
    return True


@pytest.mark.parametrize("doc", GRAMMAR, ids=[d[:24].decode("latin1") for d in GRAMMAR])
def test_grammar_numbers_atoms(doc):
    assert _check(doc)


@pytest.mark.parametrize("name", ["github_events.json", "wide_bench.json"])
def test_reference_files(name):
    assert _check(load_fixture(name).rstrip())


def test_depth_limit():
    for depth in (3, 4, 5, 10):
        for doc in (b"[[[[1]]]]", b'{"a":{"b":{"c":1}}}', b"[[[[]]]]", b"[" * 10 + b"]" * 10, b'[{"a":[{}]}]'):
            assert _check(doc, max_depth=depth)


def test_fuzz_documents():
    rng = random.Random(33)

    def value(d):
        r = rng.random()
        if d > 4 or r < 0.4:
            return rng.choice(['"s"', '"a\\nb"', '"é€"', "1", "-2.5e3", "true", "false", "null", '""', "12345678", "0.000001", "tru", "01",
                               "1.", "", "falsey", "nul", "]", "}", ":", ",", "[", "{"])
        if r < 0.7:
            return "[" + rng.choice([",", ", ", " ,", " "]).join(value(d + 1) for _ in range(rng.randint(0, 5))) + rng.choice(["]", "]", "]", "}", ""])
        return "{" + rng.choice([",", ",", " "]).join('%s%s%s' % (rng.choice(['"k%d"' % i, '"k"', "1", ""]), rng.choice([":", ":", " : ", "", ","]), value(d + 1))
                                                      for i in range(rng.randint(0, 5))) + rng.choice(["}", "}", "}", "]", ""])
    checked = 0
    for _ in range(6000):
        checked += _check(value(0).encode())
    assert checked > 5000
