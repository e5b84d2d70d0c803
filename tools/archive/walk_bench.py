#!/usr/bin/env python
"""Device-resident full parse of a batch (BASELINE.json configs[3] shape: ~1 KB documents): isolated stage 1 ->
string records -> GPU walk (tapes), each timed with events on the launch stream.
usage: walk_bench.py [n_docs] [schema]   (schema: all documents share one sequence of field types)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import simdjson_java_amd as S
import simdjson_java_amd.binding as B
if os.environ.get("SJMI_LIB"):  # A/B of library variants (tools/variants/*.so)
    B._LIB = os.environ["SJMI_LIB"]
import synth

n_want = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
same = len(sys.argv) > 2 and sys.argv[2] == "schema"
docs = synth.small_docs(n=4000, same_schema=same)
unit = b"".join(d + b"\n" for d in docs)
lens = np.array([len(d) + 1 for d in docs], dtype=np.uint64)
reps = max(1, n_want // len(docs))
n_docs, n = len(docs) * reps, len(unit) * reps
offs = np.concatenate([[0], np.cumsum(np.tile(lens, reps))]).astype(np.uint64)
buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
buf[:n] = torch.frombuffer(bytearray(unit), dtype=torch.uint8).cuda().repeat(reps)
d_offs = torch.from_numpy(offs.view(np.int64)).cuda()
d_io = torch.zeros(n_docs + 1, dtype=torch.int64, device="cuda")
d_st = torch.zeros(n_docs, dtype=torch.int32, device="cuda")
cap = n // 4 + 1
d_idx = torch.empty(cap, dtype=torch.int32, device="cuda")
res = torch.zeros(2, dtype=torch.int64, device="cuda")
ctx = S.Context(0, 1 << 20)
work = torch.cuda.Stream()  # (an explicit stream: handle 0 would mean "the context's own stream")
work.wait_stream(torch.cuda.current_stream())
torch.cuda.set_stream(work)
st = work.cuda_stream
ctx.stage1_batch_isolated_device(buf.data_ptr(), n, d_offs.data_ptr(), n_docs, d_idx.data_ptr(), cap, d_io.data_ptr(), d_st.data_ptr(), res.data_ptr(), st)
torch.cuda.synchronize()
count = int(res[0].item())
d_sb = torch.empty(n + 4 * count + 64, dtype=torch.uint8, device="cuda")
d_dso = torch.zeros(n_docs + 1, dtype=torch.int64, device="cuda")
d_ures = torch.zeros(3, dtype=torch.int64, device="cuda")
tcap = 2 * count + 2 * n_docs + 8
d_tape = torch.empty(tcap, dtype=torch.int64, device="cuda")
d_to = torch.zeros(n_docs + 1, dtype=torch.int64, device="cuda")
d_err = torch.zeros(n_docs, dtype=torch.int32, device="cuda")
d_wres = torch.zeros(4, dtype=torch.int64, device="cuda")


def stage1():
    ctx.stage1_batch_isolated_device(buf.data_ptr(), n, d_offs.data_ptr(), n_docs, d_idx.data_ptr(), cap, d_io.data_ptr(), d_st.data_ptr(), res.data_ptr(), st)


def strings():
    ctx.unescape_batch_device(buf.data_ptr(), n, d_idx.data_ptr(), count, d_offs.data_ptr(), d_io.data_ptr(), n_docs, d_sb.data_ptr(),
                              d_sb.numel(), d_dso.data_ptr(), d_ures.data_ptr(), st)


def walk():
    ctx.walk_batch_device(buf.data_ptr(), d_offs.data_ptr(), n_docs, d_idx.data_ptr(), count, d_io.data_ptr(), d_st.data_ptr(),
                          d_sb.data_ptr(), d_dso.data_ptr(), 0, 1024, d_tape.data_ptr(), tcap, d_to.data_ptr(), d_err.data_ptr(),
                          d_wres.data_ptr(), st)


ITERS = int(os.environ.get("WALK_BENCH_ITERS", "20"))  # (3 under rocprofv3 --pmc)


def timed(fn, iters=None, warm=None):
    iters = iters or ITERS
    warm = warm if warm is not None else max(1, ITERS // 4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


t1, t2, t3 = timed(stage1), timed(strings), timed(walk)
w = d_wres.cpu().numpy()
assert int(w[1]) == 0 and int(w[2]) == 0 and (int(w[3]) & 1) == 0, w
words = int(w[0])
tall = timed(lambda: (stage1(), strings(), walk()))
print("field types: %s" % ("one schema for all documents" if same else "random per field"))
print("%d documents, %d B, %d structurals, %d string bytes, %d tape words" % (n_docs, n, count, int(d_ures[0].item()), words))
print("isolated stage 1 %.3f ms | string records %.3f ms | GPU walk + pack %.3f ms | all three back to back %.3f ms = "
      "%.1f M documents/s, %.0f GB/s of JSON, device-resident in and out" % (t1, t2, t3, tall, n_docs / tall / 1e3, n / tall / 1e6))
