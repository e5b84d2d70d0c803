#!/usr/bin/env python
"""Differential soak of the round-3 string pass and of everything that feeds on it (experiments / bug hunting, not a test):
random documents heavy in escapes, \\u sequences (valid pairs, lone and reversed surrogates, bad hex), multi-byte UTF-8,
backslash runs and strings that straddle 64 B / 4 KiB / 16 KiB boundaries -- through sjmi_parse_document (all stages on the
device), the two-call host path (sjmi_stage1 + sjmi_unescape) and the drop-in parser (host walker), against the oracle.
usage: soak_strings.py <seconds> <seed>"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import simdjson_java_amd as S
from oracle import oracle as O

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
from soak_strings_gen import document

ctx = S.Context(0, 1 << 20)
par = S.SimdJsonParser(capacity=1 << 20, gpu_walk=False)
t0 = time.time()
docs = bad = 0
while time.time() - t0 < secs:
    d = document(rng)
    want = O.parse(d)
    tape, strings, err, st = ctx.parse_document(d)
    ok = st == want.stage1_status and err == want.error and (err != 0 or (np.array_equal(tape, want.tape) and strings == want.strings))
    if st == 0 and want.stage1_status == 0:  # the two-call host path: every record, failing strings marked
        idx, st2 = ctx.stage1(d)
        wsb, _, feo, fec = O.unescape_all(d + b"\0" * 64, idx)
        got, fei, gfc = ctx.unescape(len(d) + 4 * idx.size + 64)
        if feo < 0:
            ok = ok and got == wsb and fei is None
        else:
            ok = ok and fei is not None and gfc == fec
    try:  # the drop-in parser (host walker): same tape or the same exception code
        r = par.parse(d)
        ok = ok and want.error == 0 and np.array_equal(r.tape, want.tape)
    except S.JsonParsingException as e:
        ok = ok and want.error != 0
    docs += 1
    if not ok:
        bad += 1
        open(os.path.join(ROOT, "gpurun_out", "soak_bad_%d_%d.json" % (seed, docs)), "wb").write(d)
        print("MISMATCH doc", docs, len(d), "st", st, want.stage1_status, "err", err, want.error, flush=True)
        if bad > 5:
            break
print("seed %d: %d documents in %.0f s, %d bad" % (seed, docs, time.time() - t0, bad))
sys.exit(1 if bad else 0)
