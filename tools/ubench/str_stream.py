#!/usr/bin/env python
"""runs tools/ubench/str_stream.hip on twitter.json x1024 (experiment: throughput of a row-streaming string copy skeleton)"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import workloads as W
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libstrstream.so")
lib = C.CDLL(so)
lib.ub_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
doc = W.load_twitter()
buf, n = W.repeat_on_device(doc, 1024, torch.device("cuda", 0))
out = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
tot = torch.zeros(n // 4096 + 1, dtype=torch.int64, device="cuda")
work = torch.cuda.Stream(); torch.cuda.set_stream(work)
for grid in (1024, 2048, 4096, 8192):
    for _ in range(5):
        lib.ub_stream(buf.data_ptr(), n, out.data_ptr(), tot.data_ptr(), work.cuda_stream, grid)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.ub_stream(buf.data_ptr(), n, out.data_ptr(), tot.data_ptr(), work.cuda_stream, grid)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("grid %5d: %.3f ms per pass over %d MB -> %.0f GB/s of document; %d MB of string bytes staged" % (grid, ms, n // 1000000, n / ms / 1e6, int(tot.sum().item()) // 1000000))
