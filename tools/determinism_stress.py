#!/usr/bin/env python
"""The same batch again and again, every output compared BIT FOR BIT with the first run's on the device (no oracle in the loop, so
thousands of runs a minute): a fault that depends on timing and not on the input -- a race between waves, a chain that is read too
early -- shows as a run that differs from the others.  Both batch paths: the three calls (stage 1 with the optimistic plain pass ->
string pass -> walker; what tools/soak_tokens.py checks against the oracle) and the fused pipeline (BatchShard.step).  The first run
of every batch IS checked against the oracle.  usage: determinism_stress.py <seconds> <seed> [documents per batch = 1500; 0 = mixed sizes] [docgen]"""
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


class ThreeCalls:
    """gpu_walk of tests/test_gpu_walk.py with everything kept on the device"""

    def __init__(self, ctx, docs, packed=None):
        buf, offs = packed if packed is not None else _pack(docs)
        self.ctx, self.n, self.nb = ctx, len(offs) - 1, len(buf)
        self.d_buf = torch.zeros(len(buf) + 128, dtype=torch.uint8, device="cuda")
        self.d_buf[:len(buf)] = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
        self.d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
        n = self.n
        self.d_idx = torch.zeros(len(buf) + 2, dtype=torch.int32, device="cuda")
        self.d_io = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        self.d_ds = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
        self.d_res = torch.zeros(2, dtype=torch.int64, device="cuda")
        self.d_sb = torch.zeros(len(buf) + 4 * (len(buf) + 2) + 64, dtype=torch.uint8, device="cuda")
        self.d_dso = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        self.d_ures = torch.zeros(3, dtype=torch.int64, device="cuda")
        self.cap = 2 * (len(buf) + 2) + 2 * n + 8
        self.d_tape = torch.zeros(self.cap, dtype=torch.int64, device="cuda")
        self.d_to = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        self.d_err = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
        self.d_wres = torch.zeros(4, dtype=torch.int64, device="cuda")

    def run(self):
        c, st = self.ctx, torch.cuda.current_stream().cuda_stream
        c.stage1_batch_isolated_device(self.d_buf.data_ptr(), self.nb, self.d_offs.data_ptr(), self.n, self.d_idx.data_ptr(), self.d_idx.numel(),
                                       self.d_io.data_ptr(), self.d_ds.data_ptr(), self.d_res.data_ptr(), st)
        torch.cuda.synchronize()
        count = int(self.d_res[0].item())
        c.unescape_batch_device(self.d_buf.data_ptr(), self.nb, self.d_idx.data_ptr(), count, self.d_offs.data_ptr(), self.d_io.data_ptr(), self.n,
                                self.d_sb.data_ptr(), self.d_sb.numel(), self.d_dso.data_ptr(), self.d_ures.data_ptr(), st)
        c.walk_batch_device(self.d_buf.data_ptr(), self.d_offs.data_ptr(), self.n, self.d_idx.data_ptr(), count, self.d_io.data_ptr(), self.d_ds.data_ptr(),
                            self.d_sb.data_ptr(), self.d_dso.data_ptr(), 0, 1024, self.d_tape.data_ptr(), self.cap, self.d_to.data_ptr(),
                            self.d_err.data_ptr(), self.d_wres.data_ptr(), st)
        torch.cuda.synchronize()
        total, words = int(self.d_ures[0].item()), int(self.d_to[-1].item())
        return {"count": self.d_res.clone(), "idx": self.d_idx[:count + 1].clone(), "io": self.d_io.clone(), "status": self.d_ds.clone(),
                "strings": self.d_sb[:total].clone(), "dso": self.d_dso.clone(), "tape": self.d_tape[:words].clone(), "to": self.d_to.clone(),
                "errors": self.d_err.clone(), "wres": self.d_wres.clone()}


class Fused:
    def __init__(self, ctx, docs, packed=None):
        buf, offs = packed if packed is not None else _pack(docs)
        self.shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))

    def run(self):
        s = self.shard
        s.step(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        c = s.check()
        words, total = int(s.tape_offsets[-1].item()), c["string_bytes"]
        return {"idx": s.idx[:c["structurals"] + 1].clone() if "structurals" in c else s.idx[:1].clone(), "tape": s.tape[:words].clone(),
                "to": s.tape_offsets.clone(), "errors": s.doc_errors.clone(), "strings": s.sb[:total].clone(), "status": s.doc_status.clone()}


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    rng = random.Random(seed)
    ctx, ctx2 = S.Context(device=0, capacity=64 << 20), S.Context(device=0, capacity=64 << 20)
    t0 = time.time()
    runs = batches = 0
    differing = []
    docgen = len(sys.argv) > 4 and sys.argv[4] == "docgen"  # the configs[3] documents (tools/docgen.c) instead of the token documents
    first = 0
    while time.time() - t0 < secs:
        nb = n if n else rng.choice([2, 7, 64, 300, 1500])  # (0: the mixed sizes of tools/soak_tokens.py)
        packed = None
        if docgen:
            import workloads as W
            b, o = W.unique_docs(first, nb, seed=W.DOCGEN_SEED + seed)
            packed, docs, first = (bytes(b), o), None, first + nb
            if rng.random() < 0.5:  # one malformed document: the repair stage of the fused pipeline
                k = rng.randrange(nb)
                bb = bytearray(packed[0])
                bb[int(o[k]):int(o[k + 1]) - 1] = b'["unclosed'.ljust(int(o[k + 1]) - 1 - int(o[k]), b" ")
                packed = (bytes(bb), o)
        else:
            docs = [document(rng) for _ in range(nb)]
            tapes, strings, errors = gpu_walk(ctx, docs)
            check_against_oracle(docs, tapes, strings, errors)  # the inputs of this batch are right once
        for name, path in (("three calls", ThreeCalls(ctx, docs, packed)), ("fused", Fused(ctx2, docs, packed))):
            ref = path.run()
            for it in range(150 if nb >= 300 else 40):
                got = path.run()
                runs += 1
                for k in ref:
                    if got[k].shape != ref[k].shape or not torch.equal(got[k], ref[k]):
                        where = -1
                        if got[k].shape == ref[k].shape:
                            where = int(torch.nonzero(got[k] != ref[k])[0].item())
                        differing.append((batches, name, it, k, where))
                        print("seed %d batch %d %s run %d: %s differs from the first run (first at %d: %s vs %s)"
                              % (seed, batches, name, it, k, where, got[k].flatten()[where].item() if where >= 0 else "-",
                                 ref[k].flatten()[where].item() if where >= 0 else "-"), flush=True)
                if time.time() - t0 > secs:
                    break
        batches += 1
    print("seed %d: %d batches of %d documents (0 = mixed), %d repeated runs in %.0f s: %d outputs differed from their first run"
          % (seed, batches, n, runs, time.time() - t0, len(differing)))


main()
