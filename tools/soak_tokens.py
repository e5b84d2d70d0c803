#!/usr/bin/env python
"""Differential soak aimed at the TOKEN WALKER (coop_walk.hip k_tok_walk, round 5's rewrite): documents made of many small tokens
-- nesting that crosses the 64-token step boundary at every phase, empty containers and keys across it, opening brackets at the
last lane of a step, depths around the 64-level stack, 1 .. 700 tokens (the whole-document ingest up to 256 structurals, the
chunk-by-chunk one beyond) -- and every single-token mutation of them (dropped / doubled / swapped tokens and separators).  Both
the three-call path (tests' gpu_walk) and the fused pipeline (BatchShard: optimistic, exact behind a rejection) against the
oracle, document by document.  usage: soak_tokens.py <seconds> <seed>"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.setrecursionlimit(20000)
import numpy as np
import torch
import simdjson_java_amd as S
from simdjson_java_amd import sharding
from tests.test_gpu_walk import gpu_walk, check_against_oracle
from tests.test_gpu_batch import _pack

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


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    ctx = S.Context(device=0, capacity=64 << 20)
    ctx2 = S.Context(device=0, capacity=64 << 20)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    t0 = time.time()
    batches = ndocs = nbad = piped = 0
    while time.time() - t0 < secs:
        n = rng.choice([2, 7, 64, 300, 1500])
        docs = [document(rng) for _ in range(n)]
        tapes, strings, errors = gpu_walk(ctx, docs)
        check_against_oracle(docs, tapes, strings, errors)
        nbad += int(np.count_nonzero(errors))
        # the same batch through the fused pipeline
        buf, offs = _pack(docs)
        shard = sharding.BatchShard(ctx2, buf, offs, dev)
        shard.step(st)
        torch.cuda.synchronize()
        c = shard.check()
        to = shard.tape_offsets.cpu().numpy()
        tape = shard.tape.cpu().numpy().view(np.uint64)
        err = shard.doc_errors.cpu().numpy()[:n]
        assert np.array_equal(err, errors), (seed, batches)
        for k in range(n):
            if errors[k] == 0:
                assert np.array_equal(tape[int(to[k]):int(to[k]) + len(tapes[k])], tapes[k]), (seed, batches, k, docs[k][:200])
        piped += 1
        batches += 1
        ndocs += n
        del shard
    print("seed %d: %d batches, %d documents (%d failing) in %.0f s, all equal to the oracle; %d batches through the pipeline"
          % (seed, batches, ndocs, nbad, time.time() - t0, piped))


main()
