#!/usr/bin/env python
"""Differential soak of sjmi_stream_* / sjmi_split_* (experiments / bug hunting): random documents (escape-heavy strings, long
backslash runs, lone quotes, broken UTF-8) cut at random multiples of 64 bytes -- fed chunk by chunk through one stream, and as
shards of virtual ranks through the split protocol -- against stage 1 of the whole document by the oracle.
usage: soak_stream.py <seconds> <seed>"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import simdjson_java_amd as S
from oracle import oracle as O
import soak_strings_gen as G

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
ctx = S.Context(0, 1 << 22)
dev = torch.device("cuda", 0)
t0 = time.time()
docs = halo_errors = 0
while time.time() - t0 < secs:
    d = bytearray(G.document(rng, rng.choice([300, 2000, 9000, 40000, 200000])))
    r = rng.random()
    if r < 0.2:
        for _ in range(rng.randint(1, 5)):
            d[rng.randrange(len(d))] = rng.choice([0x22, 0x5C, 0x01, 0xFF, 0xC3])
    elif r < 0.3:
        p = rng.randrange(len(d))
        d[p:p] = b"\\" * rng.choice([63, 64, 65, 127, 128, 129, 500, 5000])
    d = bytes(d)
    want_idx, want_st = O.stage1(d)
    # ---- stream: random chunking (every chunk but the last a non-zero multiple of 64)
    cuts = sorted(set(rng.randrange(64, max(len(d), 65), 64) for _ in range(rng.randint(0, 6))) | {0})
    cuts = [c for c in cuts if c < len(d)] + [len(d)]
    s = ctx.stream(max(b - a for a, b in zip(cuts, cuts[1:])) + 64)
    got, st = [], 0
    try:
        for a, b in zip(cuts, cuts[1:]):
            base, idx, st = s.push(d[a:b], b == len(d))
            assert base == a
            got.append(idx.astype(np.int64) + base)
        ok = st == want_st and np.array_equal(np.concatenate(got) if got else np.zeros(0, np.int64), want_idx.astype(np.int64))
    except S.SjmiError:
        ok = b"\\" * 4000 in d  # (a backslash run longer than everything the stream keeps: reported, not guessed)
        halo_errors += 1
    finally:
        s.close()
    # ---- split: the same cuts as shards of virtual ranks
    if ok and len(cuts) > 2:
        bufs, idxs, sp = [], [], []
        halo = 64 if b"\\" * 60 not in d else 8192
        for rk, (a, b) in enumerate(zip(cuts, cuts[1:])):
            h = min(halo, a) // 64 * 64
            t = torch.zeros(h + (b - a) + 128, dtype=torch.uint8, device=dev)
            t[:h + b - a] = torch.frombuffer(bytearray(d[a - h:b]), dtype=torch.uint8).to(dev)
            ix = torch.empty(b - a + 66, dtype=torch.int32, device=dev)
            bufs.append(t); idxs.append(ix)
            sp.append(ctx.split(t.data_ptr() + h, b - a, h, h == a, rk == len(cuts) - 2, ix.data_ptr(), ix.numel()))
        flips = [x.scan()[0] for x in sp]
        out, status, after = [], 0, 0
        for rk, x in enumerate(sp):
            count, st2, after = x.resolve(sum(flips[:rk]) & 1)
            status |= st2
            out.append(idxs[rk][:count].cpu().numpy().view(np.uint32).astype(np.int64) + cuts[rk])
            x.close()
        if after:
            status |= O.ST_UNCLOSED
        ok = (status & 0xFF) == want_st and np.array_equal(np.concatenate(out), want_idx.astype(np.int64))
    docs += 1
    if not ok:
        open(os.path.join(ROOT, "gpurun_out", "soak_stream_bad_%d_%d.json" % (seed, docs)), "wb").write(d)
        print("MISMATCH doc", docs, len(d), cuts, flush=True)
        sys.exit(1)
print("seed %d: %d documents in %.0f s (stream + split) equal to the whole-document oracle; %d over-long runs reported" % (seed, docs, time.time() - t0, halo_errors))
