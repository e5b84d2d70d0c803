"""The algebra of unescape_packed (csrc/unescape.hip) -- many escaped strings as one packed byte stream -- restated in
tools/packed_unescape_model.py and checked against the oracle's per-string StringParser restatement: random token soups so
that string boundaries, escapes, \\uXXXX and surrogate pairs fall on every window position, strings ending in backslash pairs
in front of strings beginning with them, and every StringParser error inside one of many strings."""
import os
import random
import sys

import numpy as np

from oracle import oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import packed_unescape_model as M  # noqa: E402

TOKS = ["a", "b", " ", "é", "€", "\\n", "\\t", "\\\\", "\\\"", "\\/", "\\u0041", "\\u00e9", "\\u20AC", "\\uD83D\\uDE00", "\\\\\\\\", "\\u0000",
        "\\uDBFF\\uDFFF"]
BAD = ["\\uD83Dx", "\\uD83D\\u0041", "\\uDE00", "\\u12G4", "\\q", "\\uD83D\\uD83D", "\\uD83D"]


def _oracle_strings(doc):
    """-> [(open, close, bytes or -code)] for every string of the document, by the oracle"""
    idx, st = O.stage1(doc)
    assert st == 0
    padded = doc + b"\0" * 64
    out = []
    for i, p in enumerate(idx):
        if doc[p] != 0x22:
            continue
        close = p + 1
        while True:  # the closing quote = the first unescaped quote
            if padded[close] == 0x5C:
                close += 2
            elif padded[close] == 0x22:
                break
            else:
                close += 1
        sb, _, feo, fec = O.unescape_all(padded, np.array([p], dtype=np.uint32))
        out.append((int(p), close, -fec if feo >= 0 else sb[4:]))
    return out


def test_packed_stream_equals_per_string_unescape():
    rng = random.Random(808)
    checked = errors = 0
    for it in range(250):
        parts = []
        for _ in range(rng.randint(1, 40)):
            n, s, size = rng.randint(1, rng.choice([4, 20, 70, 200])), [], 0
            while size < n:
                t = rng.choice(BAD) if rng.random() < 0.01 else rng.choice(TOKS)
                s.append(t)
                size += len(t.encode())
            parts.append('"%s"' % "".join(s))
        doc = ("[" + rng.choice([",", ", ", ",\n"]).join(parts) + "]").encode()
        strs = _oracle_strings(doc)
        esc = [(o, c, w) for (o, c, w) in strs if b"\\" in doc[o:c] and c - o - 1 <= 256]
        got = M.unescape_packed(doc + b"\0" * 64, [(o, c) for o, c, _ in esc])
        for (o, c, want), g in zip(esc, got):
            assert g == want, (it, doc[o:c + 1], g, want)
            checked += 1
            errors += isinstance(want, int)
    assert checked > 3000 and errors > 20


def test_backslash_runs_touching_across_string_boundaries():
    doc = ("[" + ",".join('"x%s\\\\"' % ("y" * k) for k in range(70)) + "," + ",".join('"\\\\%s\\\\"' % ("z" * k) for k in range(70)) + "]").encode()
    strs = _oracle_strings(doc)
    got = M.unescape_packed(doc + b"\0" * 64, [(o, c) for o, c, _ in strs])
    assert got == [w for _, _, w in strs]
