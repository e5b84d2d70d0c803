#!/usr/bin/env python
"""Differential soak of the BATCH paths (experiments / bug hunting): random batches of small documents -- escape-heavy strings,
grammar errors, documents that fail stage 1 (unclosed strings, bad UTF-8, control characters: the sanitised-copy string pass) --
through stage1_batch_isolated_device -> unescape_batch_device -> walk_batch_device (tests' gpu_walk) against the oracle,
document by document.  usage: soak_batch.py <seconds> <seed>"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.setrecursionlimit(20000)
import simdjson_java_amd as S
from tests.test_gpu_walk import gpu_walk, check_against_oracle
from tests.test_gpu_coop_walk import _adversarial
import soak_strings_gen as G

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
ctx = S.Context(device=0, capacity=64 << 20)
t0 = time.time()
batches = ndocs = 0
while time.time() - t0 < secs:
    n = rng.choice([1, 2, 7, 64, 300, 2000])
    docs = []
    for _ in range(n):
        r = rng.random()
        if r < 0.5:
            d = G.document(rng, rng.choice([60, 200, 900, 4100, 9000]))
        elif r < 0.8:
            d = _adversarial(rng, 1)[0]
        elif r < 0.9:
            d = G.document(rng, 300)
            cut = rng.randrange(1, len(d))
            d = d[:cut]                                  # truncated: unclosed strings / containers
        else:
            d = bytearray(G.document(rng, 300))
            d[rng.randrange(len(d))] = rng.choice([0x01, 0x22, 0x5C, 0xFF, 0xC3, 0x0A])
            d = bytes(d)
        docs.append(d)
    clean = rng.random() < 0.3
    if clean:  # batches the optimistic plain pass accepts: valid documents only
        docs = [G.document(rng, rng.choice([60, 200, 900])) for _ in range(n)]
    tapes, strings, errors = gpu_walk(ctx, docs)
    check_against_oracle(docs, tapes, strings, errors)
    batches += 1
    ndocs += n
print("seed %d: %d batches, %d documents in %.0f s, all equal to the oracle" % (seed, batches, ndocs, time.time() - t0))
