#!/usr/bin/env python
"""Differential soak aimed at the TOKEN WALKER (coop_walk.hip k_tok_stream; written for round 5's k_tok_walk): documents made of many small tokens
-- nesting that crosses the 64-token step boundary at every phase, empty containers and keys across it, opening brackets at the
last lane of a step, depths around the 64-level stack, 1 .. 700 tokens (the whole-document ingest up to 256 structurals, the
chunk-by-chunk one beyond) -- and every single-token mutation of them (dropped / doubled / swapped tokens and separators).  Both
the three-call path (tests' gpu_walk) and the fused pipeline (BatchShard: optimistic, exact behind a rejection) against the
oracle, document by document.  usage: soak_tokens.py <seconds> <seed> [documents per batch]"""
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

from token_docs import document  # noqa: E402


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
        n = int(sys.argv[3]) if len(sys.argv) > 3 else rng.choice([2, 7, 64, 300, 1500])
        docs = [document(rng) for _ in range(n)]
        tapes, strings, errors = gpu_walk(ctx, docs)
        try:
            check_against_oracle(docs, tapes, strings, errors)
        except AssertionError as e:  # what differs, for the record: the document, its neighbours, the first differing tape word
            from oracle import oracle as O
            k = e.args[0] if isinstance(e.args[0], int) else e.args[0][0]
            want = O.parse(docs[k] + b"\n")
            print("seed %d batch %d (n = %d): document %d differs" % (seed, batches, n, k))
            for j in range(max(0, k - 2), min(n, k + 3)):
                print("  doc %d (%d bytes, error %d): %r" % (j, len(docs[j]), int(errors[j]), docs[j][:400]))
            if tapes[k] is not None and not want.error:
                got, exp = tapes[k], want.tape
                print("  tape words: got %d, want %d" % (got.size, exp.size))
                for i in range(min(got.size, exp.size)):
                    if got[i] != exp[i] and (int(exp[i]) >> 56) != ord('"'):
                        print("  first differing non-string word %d: got %016x want %016x" % (i, int(got[i]), int(exp[i])))
                        break
                print("  got  %r" % (O.Parsed(got, strings, 0, 0, 0).to_python(),))
                print("  want %r" % (want.to_python(),))
            for again in range(3):  # the same batch again: a deterministic fault shows every time
                try:
                    t2, s2, e2 = gpu_walk(ctx, docs)
                    check_against_oracle(docs, t2, s2, e2)
                    print("  the same batch again (%d): equal to the oracle" % again)
                except AssertionError as e2x:
                    print("  the same batch again (%d): differs again: %r" % (again, e2x.args[:1]))
            raise
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
