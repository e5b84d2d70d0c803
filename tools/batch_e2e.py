#!/usr/bin/env python
"""End-to-end batched parse (BASELINE configs[3]/[4] style, scaled): documents in host memory -> tapes in host memory.
Times the C call sjmi_parser_parse_batch only (no Python-side copies)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import simdjson_java_amd as S
import simdjson_java_amd.binding as B
import synth
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
docs = synth.small_docs(n=4000)
unit = b"".join(d + b"\n" for d in docs)
lens = np.array([len(d) + 1 for d in docs], dtype=np.uint64)
reps = n_docs // len(docs)
buf = np.frombuffer(unit * reps, dtype=np.uint8)
offs = np.concatenate([[0], np.cumsum(np.tile(lens, reps))]).astype(np.uint64)
n = offs.size - 1
L = B.lib()
tape_p, to_p, sb_p, err_p = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint8)(), C.POINTER(C.c_int32)()
sb_len = C.c_uint64(0)
for threads in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "4", "16", "32", "64"]):
    os.environ["SJMI_PARSE_THREADS"] = threads  # read when the parser is created
    p = S.SimdJsonParser(capacity=buf.size + 64)

    def call():
        rc = L.sjmi_parser_parse_batch(p._h, buf.ctypes.data, buf.size, offs.ctypes.data, n, C.byref(tape_p), C.byref(to_p),
                                       C.byref(sb_p), C.byref(sb_len), C.byref(err_p))
        assert rc == 0, rc
    call()
    call()
    t0 = time.perf_counter()
    for _ in range(5):
        call()
    t = (time.perf_counter() - t0) / 5
    errs = np.ctypeslib.as_array(err_p, shape=(n,))
    print("parse_batch, %2s host threads: %d documents, %d MB: %.1f ms -> %.2f M docs/s, %.2f GB/s end to end (host in, "
          "tapes out); errors: %d" % (threads, n, buf.size // 1000000, t * 1e3, n / t / 1e6, buf.size / t / 1e9,
                                     int((errs != 0).sum())))
    p.close()
