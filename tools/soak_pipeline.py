"""Differential soak of the device pipeline (isolated stage 1 -> string records -> cooperative walker) against the oracle:
seeded random corpora far larger than the test-suite's -- adversarial grammar, escape soups (the packed stream of
unescape.hip), number literals around rounding boundaries, nested documents -- as batches and, for a sample, as single
documents through sjmi_parse_document (chunk-parallel walker for the large ones).  Run on the GPU box:
    python tools/soak_pipeline.py [rounds] [docs per round] [seed base]"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simdjson_java_amd as S  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.test_gpu_coop_walk import _adversarial  # noqa: E402
from tests.test_gpu_walk import check_against_oracle, gpu_walk  # noqa: E402
from tests.walk_common import NEEDS_HOST, exact_range, random_number_literal  # noqa: E402

TOKS = ["a", "b", "Z", " ", "é", "€", "\\n", "\\t", "\\\\", "\\\"", "\\/", "\\b", "\\u0041", "\\u00e9", "\\u20AC", "\\uD83D\\uDE00",
        "\\\\\\\\", "\\u0000", "\\uFFFF", "\\uDBFF\\uDFFF"]
BAD = ["\\uD83Dx", "\\uDE00", "\\u12G4", "\\q", "\\uD83D\\uD83D", "\\u"]


def soup(rng, n, p_bad):
    out, size = [], 0
    while size < n:
        t = rng.choice(BAD) if rng.random() < p_bad else rng.choice(TOKS)
        out.append(t)
        size += len(t.encode())
    return "".join(out)


def value(rng, d):
    r = rng.random()
    if d > 5 or r < 0.35:
        k = rng.random()
        if k < 0.4:
            return '"%s"' % soup(rng, rng.randint(0, rng.choice([4, 20, 60, 300])), 0.002)
        if k < 0.7:
            return random_number_literal(rng)
        return rng.choice(["true", "false", "null", "[]", "{}", "0", "-1"])
    if r < 0.7:
        return "[" + ",".join(value(rng, d + 1) for _ in range(rng.randint(0, 6))) + "]"
    return "{" + ",".join('"%s":%s' % (soup(rng, rng.randint(1, 12), 0.001), value(rng, d + 1)) for _ in range(rng.randint(0, 6))) + "}"


def fused_check(ctx, docs, want_err, want_tapes, stats):
    import torch
    from simdjson_java_amd import sharding
    buf = b"".join(d + b"\n" for d in docs)
    offs = np.concatenate([[0], np.cumsum([len(d) + 1 for d in docs])]).astype(np.uint64)
    shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
    shard.step(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    shard.check()
    to = shard.tape_offsets.cpu().numpy()
    tape = shard.tape.cpu().numpy().view(np.uint64)
    err = shard.doc_errors.cpu().numpy()[:len(docs)]
    assert np.array_equal(err, np.asarray(want_err, dtype=err.dtype)), "fused pipeline: document verdicts differ"
    for k in range(len(docs)):
        if err[k] == 0:
            got = tape[int(to[k]):int(to[k + 1])]
            assert got.size == want_tapes[k].size, k
            # (STRING payloads are offsets into the batch's own string buffer: compare everything else word for word)
            strings_at = (want_tapes[k] >> np.uint64(56)) == np.uint64(ord('"'))
            assert np.array_equal(got[~strings_at], want_tapes[k][~strings_at]), k
    stats["fused"] = stats.get("fused", 0) + len(docs)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 7000
    ctx = S.Context(device=0, capacity=256 << 20)
    stats = {"docs": 0, "ok": 0, "errors": 0, "host": 0, "single": 0}
    for rnd in range(rounds):
        rng = random.Random(seed0 + rnd)
        docs = _adversarial(rng, n // 4) + [value(rng, 0).encode() for _ in range(3 * n // 4)]
        rng.shuffle(docs)
        docs = [d for d in docs if b"\n" not in d or True]
        tapes, strings, errors = gpu_walk(ctx, docs)
        host_ok = set(k for k in range(len(docs)) if int(errors[k]) == NEEDS_HOST)
        # (a hand-back wins over an error BEHIND it -- the host walks the document again for the exact verdict; the soak only
        #  counts them: the generators aim at rounding boundaries on purpose)
        check_against_oracle(docs, tapes, strings, errors, host_ok)
        stats["docs"] += len(docs)
        stats["ok"] += int((errors == 0).sum())
        stats["errors"] += int((errors > 0).sum())
        stats["host"] += len(host_ok)
        # the fused pipeline (sjmi_parse_batch_device): the mixed batch (per-document passes) and the valid documents alone
        # (newline-separated and clean: the optimistic plain stage 1) must reproduce the three separate calls
        good = [d for k, d in enumerate(docs) if int(errors[k]) == 0]
        good_tapes = [tapes[k] for k in range(len(docs)) if int(errors[k]) == 0]
        for batch, want_err, want_tapes in ((docs, errors, tapes), (good, np.zeros(len(good), dtype=np.int32), good_tapes)):
            fused_check(ctx, batch, want_err, want_tapes, stats)
        # single documents: a sample of the small ones, and everything concatenated into large arrays (chunk-parallel)
        big = [b"[" + b",".join(good[i::7]) + b"]" for i in range(7)]
        bad = [b"[" + b",".join(docs[i:i + 400]) + b"]" for i in range(0, min(len(docs), 4000), 400)]
        for d in rng.sample(docs, 300) + big + bad:
            want = O.parse(d)
            tape, sb, err, st = ctx.parse_document(d)
            assert st == want.stage1_status, d[:60]
            if err == NEEDS_HOST:
                continue
            assert err == want.error, (d[:80], err, want.error)
            if err == 0:
                assert np.array_equal(tape, want.tape) and sb == want.strings, d[:80]
            stats["single"] += 1
        print("round", rnd, stats, flush=True)
    ctx.close()
    print("soak ok", stats)


if __name__ == "__main__":
    main()
