"""Token-level adversarial documents for the batch walker (coop_walk.hip k_tok_stream): documents made of many small tokens -- nesting
that crosses the 64-token step boundary at every phase, empty containers and keys across it, opening brackets at the last lane of
a step, depths around the 64-level stack, 1 .. 700 tokens -- and single-token mutations of them (dropped / doubled / swapped tokens
and separators).  Pure Python: used by tools/soak_tokens.py (GPU) and tests/test_tok_walk_model.py (CPU)."""
SCALARS = [b"1", b"-7", b"0", b"12345678", b"123456789012345", b"1234567890123456", b"1.5", b"1e3", b"true", b"false", b"null",
           b'"a"', b'""', b'"k\\n"', b'"\\u00e9"', b"[]", b"{}", b"[[]]", b'{"a":{}}']
BAD_SCALARS = [b"tru", b"nul", b"01", b"-", b"1.", b'"\\q"', b"falsee", b"1e", b"+1"]


def value(rng, budget, depth, max_depth):
    """-> (list of tokens incl. separators, tokens used)"""
    if budget <= 1 or depth >= max_depth or rng.random() < 0.35:
        return [rng.choice(SCALARS)], 1
    arr = rng.random() < 0.5
    out = [b"[" if arr else b"{"]
    used = 1
    n = rng.choice([0, 1, 1, 2, 3, 5, 9, 17, 40])
    for i in range(n):
        if used >= budget:
            break
        if i:
            out.append(b",")
        if not arr:
            out += [b'"k%d"' % i, b":"]
            used += 1
        sub, u = value(rng, budget - used, depth + 1, max_depth)
        out += sub
        used += u
    out.append(b"]" if arr else b"}")
    return out, used + 1


def deep(rng, d, filler):
    """d levels of nesting with `filler` scalars in front of the innermost container (the step boundary moves through the levels)"""
    out = []
    for i in range(d):
        out.append(b"[" if rng.random() < 0.5 else b'{"k":')
        if out[-1] == b"[" and i < filler:
            out += [rng.choice(SCALARS[:6]), b","]
    out.append(rng.choice(SCALARS))
    for t in reversed([x for x in out[:-1] if x in (b"[", b'{"k":')]):
        out.append(b"]" if t == b"[" else b"}")
    return out


def mutate(rng, toks):
    t = list(toks)
    if not t:
        return t
    i = rng.randrange(len(t))
    k = rng.randrange(7)
    if k == 0:
        del t[i]
    elif k == 1:
        t.insert(i, t[i])
    elif k == 2 and len(t) > 1:
        j = rng.randrange(len(t))
        t[i], t[j] = t[j], t[i]
    elif k == 3:
        t[i] = rng.choice([b",", b":", b"[", b"]", b"{", b"}"])
    elif k == 4:
        t[i] = rng.choice(BAD_SCALARS)
    elif k == 5:
        t.insert(i, rng.choice([b",", b":"]))
    else:
        t = t[:i]
    return t


def document(rng):
    r = rng.random()
    if r < 0.55:
        toks, _ = value(rng, rng.choice([3, 20, 60, 64, 65, 70, 127, 128, 129, 200, 400, 700]), 0, rng.choice([3, 6, 12, 70]))
        if toks[0] not in (b"[", b"{"):
            toks = [b"["] + toks + [b"]"]
    elif r < 0.75:
        toks = deep(rng, rng.choice([5, 30, 62, 63, 64, 65, 66, 70]), rng.choice([0, 1, 3, 20, 61, 62, 63, 64]))
    else:
        # a long flat array / object: the boundary falls on every kind of token sooner or later
        n = rng.choice([31, 32, 33, 63, 64, 65, 127, 128, 129, 300])
        if rng.random() < 0.5:
            toks = [b"["] + sum(([rng.choice(SCALARS), b","] for _ in range(n)), [])[:-1] + [b"]"]
        else:
            toks = [b"{"] + sum(([b'"k"', b":", rng.choice(SCALARS), b","] for _ in range(n)), [])[:-1] + [b"}"]
    if rng.random() < 0.45:
        toks = mutate(rng, toks)
    sep = rng.choice([b"", b" ", b"", b"  "])
    return sep.join(toks) or b"[]"


