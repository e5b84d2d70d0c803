#!/usr/bin/env python
"""bench.py -- stage-1 throughput of the MI355X engine on BASELINE.json's metric.

One "step" = one full pass of the hot path (fused UTF-8 validation + structural indexing + index
compaction, i.e. SimdJsonParser.stage1) over one batch of input that is already resident in HBM:
twitter.json x 1024 byte-concatenated (646,671,360 B; BASELINE.json configs[1]).  With --gpus N the
batch is sharded by document (every rank scans its own 1024 copies: weak scaling) and RCCL is used
only to all-gather the per-shard {count, status} records, as north_star prescribes.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (k_stage1) vs the 8 TB/s HBM peak, timed with HIP events attached to
                  the kernel's dispatch on its launch stream (sjmi_set_profiling)
  cpu_baseline -- the C oracle (a port of the reference's Java stage 1) on all host cores, on a
                  bounded sample of the same workload
"""
import argparse
import gzip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def load_twitter():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "data", "twitter.json.gz"), "rb") as f:
        return f.read()


def cpu_baseline(doc, seconds=10.0):
    """Oracle (C port of StructuralIndexer.index512 + Utf8Validator.validate) on ALL host cores: one thread per core,
    each scanning its own copy of the sample (the C call releases the GIL); plus the single-core rate for reference."""
    import threading
    import numpy as np
    from oracle import oracle
    oracle.build()
    reps = 16
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    base = np.frombuffer(doc * reps, dtype=np.uint8)
    oracle.stage1(base)  # warm
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < 2.0:  # single core
        oracle.stage1(base)
        n1 += base.size
    one = n1 / (time.perf_counter() - t0) / 1e9
    samples = [base.copy() for _ in range(cores)]
    done = [0] * cores
    stop = time.perf_counter() + seconds

    def work(k):
        while time.perf_counter() < stop:
            oracle.stage1(samples[k])
            done[k] += samples[k].size

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    total = sum(done)
    return {"value": round(total / el / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": "twitter.json x%d (%d B) per thread, %d threads, %d scans in %.1f s by oracle/sj_oracle.c (scalar C "
                      "restatement of the reference's Java stage 1; the reference itself needs a JVM, absent here); one "
                      "core alone: %.3f GB/s" % (reps, base.size, cores, total // base.size, el, one)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--preheat", type=int, default=400,
                    help="untimed launches before the warmup steps: the GPU's clock governor needs ~100 back-to-back "
                         "launches (25 ms) of this kernel to settle (per-launch times: tools/perlaunch.py)")
    ap.add_argument("--reps", type=int, default=1024, help="copies of twitter.json per GPU (1024 = configs[1])")
    ap.add_argument("--tile-steps", type=int, default=0, help="force the chain granule = N x 4 KiB: 1, 2 or 4 (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import simdjson_java_amd as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    doc = load_twitter()
    n0 = len(doc)
    reps = args.reps
    n = n0 * reps
    assert n < (1 << 32)
    from oracle import oracle  # checker only: expected indexes of ONE copy
    oracle.build()
    idx0, st0 = oracle.stage1(doc)
    s_total = idx0.size * reps

    dev = torch.device("cuda", local_rank)
    buf = torch.zeros(n + 128, dtype=torch.uint8, device=dev)
    buf[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).to(dev).repeat(reps)
    cap = s_total + 1
    out = torch.empty(cap, dtype=torch.int32, device=dev)
    res = torch.zeros(2, dtype=torch.int64, device=dev)
    gathered = torch.zeros(2 * world, dtype=torch.int64, device=dev) if world > 1 else None

    ctx = S.Context(device=local_rank, capacity=1 << 20)
    if args.tile_steps:
        ctx.set_tile_steps(args.tile_steps)
    assert ctx.selftest() == 0
    # one explicit (non-default) stream for kernels, copies and the collective, so that everything is ordered
    work = torch.cuda.Stream(device=dev)
    work.wait_stream(torch.cuda.current_stream())
    stream = work.cuda_stream

    def step():
        with torch.cuda.stream(work):
            ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), stream)
            if world > 1:
                dist.all_gather_into_tensor(gathered, res)  # per-shard {count, status}: the only collective

    step()
    torch.cuda.synchronize()
    # parity check outside the timed region: closed form index[k*S+j] = k*N0 + index0[j]
    r = res.cpu().numpy()
    assert int(r[0]) == s_total and (int(r[1]) & 0xFFFFFFFF) == 0, r
    want = (torch.from_numpy(idx0.astype(np.int64)).to(dev)[None, :] +
            (torch.arange(reps, device=dev, dtype=torch.int64) * n0)[:, None]).flatten()
    got = out[:s_total].to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(got, want), "GPU indexes differ from the oracle's closed form"
    assert int(out[s_total].item()) == 0
    del want, got

    for _ in range(args.preheat + args.warmup):  # untimed
        step()
    ctx.set_profiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms, launches = ctx.kernel_time()
    ctx.set_profiling(False)

    if rank == 0:
        # HBM traffic per launch from the committed PMC passes of this same command (rocprofv3 cannot run inside
        # the timed region): profiles/r1/pmc_summary.json, FETCH_SIZE doubled per the gfx950 note. null if absent
        # or if the workload differs from the profiled one.
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r1", "pmc_summary.json")) as f:
                if reps == 1024:
                    traffic = int(json.load(f)["hbm_traffic_bytes_per_launch"]["total"])
        except (OSError, KeyError, ValueError):
            traffic = None
        ms_per_step = elapsed / args.steps * 1e3
        value = n * world * args.steps / elapsed / 1e9
        avg_kernel_s = kern_ms / max(launches, 1) / 1e3
        achieved = n / avg_kernel_s / 1e9
        line = {
            "metric": "GB/s JSON scanned (stage-1)", "value": round(value, 2), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic: twitter.json (631,515 B reference fixture) x%d byte-concatenated per GPU, resident in HBM" % reps,
            "config": {"workload": "twitter.json x%d concatenated: stage-1 = UTF-8 validation + structural indexing + "
                                   "uint32 index compaction, one fused single-pass kernel, bit-exact index check" % reps,
                       "bytes_per_gpu": n, "structurals_per_gpu": s_total, "tile_steps": args.tile_steps or "auto",
                       "preheat_launches": args.preheat,
                       "sharding": "by document, RCCL all_gather of per-shard {count,status} only" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "k_stage1", "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": launches,
                         "algorithmic_bytes_per_launch": n,
                         "achieved_incl_index_writes": round((n + 4 * (s_total + 1)) / avg_kernel_s / 1e9, 2)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(doc)
        print(json.dumps(line))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
