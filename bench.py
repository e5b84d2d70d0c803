#!/usr/bin/env python
"""bench.py -- the MI355X engine on BASELINE.json's metrics.

--gpus 1 (default; the driver's BENCH line)
    One "step" = one pass of the hot path (fused UTF-8 validation + structural indexing + index compaction, i.e.
    SimdJsonParser.stage1) over the north-star workload, resident in HBM: twitter.json x 6801 byte-concatenated =
    4,294,933,515 B ("4 GiB concatenated twitter.json"), checked bit-exactly against the oracle's closed form on the
    device before anything is timed.  `value` = GB/s of JSON over the K timed steps.  The `roofline` object is about
    the dominant kernel (k_stage1) and gives BOTH its cold time (the first K launches after the GPU sat idle) and its
    clock-settled time (`frac` is the settled one; `cold` holds the other).  `extra` carries the other configs of
    BASELINE.json measured in the same run, each with its own roofline block: configs[1] (twitter x1024), configs[2]
    (4 GiB synthetic), the string-unescape path on twitter x1024, and configs[3] (1,000,000 UNIQUE ~1 KB documents, tools/docgen.c:
    stage 1 with per-document verdicts -> string records -> GPU walk; a seeded sample of 10,000 checked against the oracle before
    timing) in documents/s, kernel-only and including the H2D copy; their headline figures are repeated as flat scalars in `config`.
--gpus N > 1 (the driver's SCALE lines; one rank per GPU under torch.distributed.run -- launched plainly as
    `python bench.py --gpus N` it re-executes itself that way; fewer than N visible GPUs, or a process group whose size is
    not N, is an ERROR (exit 3), never a silent single-GPU run)
    The batched mode, sharded by document, RCCL ONLY as the all_gather of 4 x int64 per rank.  `metric` / `value` are the
    N = 1 line's, weak-scaled: every rank scans its own 6801 twitter.json documents (4 GiB) per step, then the count
    gather; value = aggregate GB/s, so value(N) / (N x value(1)) is the efficiency of the headline metric.  `batched` holds
    configs[3] in documents/s both ways -- weak (`--docs` unique ~1 KB documents per rank) and strong (`--docs` in all,
    contiguous byte-balanced ranges) -- each with documents_per_rank and the RCCL world size it saw, beside rank 0
    running `--docs` documents alone in the same run (`single_gpu_same_run`).

Prints ONE JSON line on rank 0.  cpu_baseline (N=1 only) = the AVX-512 restatement of the reference's two stage-1
passes (oracle/sj_avx512.c) on 1 core and on all host cores; the reference itself is Java and needs a JVM, which is
probed for and reported."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s achievable)
PROFILE_ROUND = "r6"
PROFILE_DIR = os.path.join(ROOT, "profiles", PROFILE_ROUND)


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------------
JAVA_HARNESS = os.path.join(ROOT, "java", "bench", "org", "simdjson", "RefStage1Bench.java")
REFERENCE_SRC_CANDIDATES = (os.environ.get("SJMI_REFERENCE_SRC", ""), "/root/reference/src/main/java")
JAVA_MIN_MAJOR = int(os.environ.get("SJMI_JAVA_MIN", "24"))  # the reference's README.md:76 asks for JDK 24


def java_major(exe):
    """major version of a `java` / `javac` executable, or None"""
    import re
    try:
        out = subprocess.run([exe, "-version"], capture_output=True, text=True, timeout=20)
    except Exception:  # noqa: BLE001
        return None
    m = re.search(r'(?:version "|javac )(\d+)(?:\.(\d+))?', (out.stderr or "") + (out.stdout or ""))
    if not m:
        return None
    major = int(m.group(1))
    return int(m.group(2) or 0) if major == 1 else major  # ("1.8.0" -> 8)


def probe_java():
    """Is a JDK that could run the reference (JDK >= 24 with jdk.incubator.vector) on this box?  (The reference's sources
    are not on the GPU box in any case: /root/reference exists only in the build container.)"""
    exe = shutil.which("java")
    if not exe:
        return "no java on PATH"
    try:
        out = subprocess.run([exe, "-version"], capture_output=True, text=True, timeout=20)
        return "java present: " + (out.stderr or out.stdout).splitlines()[0]
    except Exception as e:  # noqa: BLE001
        return "java probe failed: %r" % (e,)


def reference_jvm_commands(ref_src, out_dir, doc_path, threads, seconds, slice_bytes=32 << 20, species="512"):
    """-> (javac command, java command) that build the reference's main sources + java/bench/org/simdjson/RefStage1Bench.java
    into out_dir and run the harness (plain javac, no Gradle, no JMH; src/main has no external dependency)"""
    sources = sorted(os.path.join(d, f) for d, _, fs in os.walk(ref_src) for f in fs if f.endswith(".java"))
    javac = [shutil.which("javac") or "javac", "--add-modules", "jdk.incubator.vector", "-nowarn", "-d", out_dir] + sources + [JAVA_HARNESS]
    java = [shutil.which("java") or "java", "--add-modules", "jdk.incubator.vector", "-Dorg.simdjson.species=%s" % species,
            "-cp", out_dir, "org.simdjson.RefStage1Bench", doc_path, str(int(threads)), "%.1f" % seconds, str(int(slice_bytes))]
    return javac, java


def reference_jvm_baseline(doc, seconds=5.0):
    """The REFERENCE ITSELF on this box's host cores -- Utf8Validator.validate + StructuralIndexer.index (= SimdJsonParser.stage1,
    SimdJsonParser.java:55-58) and SimdJsonParser.parse, 1 thread and all cores -- when a JDK >= 24 (java + javac) and the
    reference's sources are here.  -> (dict or None, reason).  Never raises: the port stays the baseline when it cannot run."""
    import tempfile
    java, javac = shutil.which("java"), shutil.which("javac")
    if not java or not javac:
        return None, "no JDK on PATH (java: %s, javac: %s)" % (java or "absent", javac or "absent")
    major = java_major(java)
    if major is None or major < JAVA_MIN_MAJOR:
        return None, "JDK %s < %d" % (major, JAVA_MIN_MAJOR)
    ref = next((d for d in REFERENCE_SRC_CANDIDATES if d and os.path.isdir(os.path.join(d, "org", "simdjson"))), None)
    if not ref:
        return None, "JDK %d present, but the reference's sources are not on this box (SJMI_REFERENCE_SRC)" % major
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with tempfile.TemporaryDirectory(prefix="sjmi_ref_") as tmp:
            doc_path = os.path.join(tmp, "doc.json")
            with open(doc_path, "wb") as f:
                f.write(doc)
            out_dir = os.path.join(tmp, "classes")
            os.makedirs(out_dir)
            c_javac, c_java = reference_jvm_commands(ref, out_dir, doc_path, cores, seconds)
            b = subprocess.run(c_javac, capture_output=True, text=True, timeout=300)
            if b.returncode != 0:
                return None, "javac failed: " + (b.stderr or b.stdout)[-300:]
            r = subprocess.run(c_java, capture_output=True, text=True, timeout=120 + 6 * seconds)
            if r.returncode != 0:
                return None, "java failed: " + (r.stderr or r.stdout)[-300:]
            res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            res["cores"] = cores
            res["command"] = " ".join(c_java[:5]) + " ... RefStage1Bench"
            return res, "ok"
    except Exception as e:  # noqa: BLE001
        return None, "reference harness failed: %r" % (e,)


def cpu_baseline(doc, seconds=8.0):
    """The reference's stage 1 (Utf8Validator.validate + StructuralIndexer.index at 512 bits) restated with AVX-512
    intrinsics (oracle/sj_avx512.c), one thread per host core, each over its own copy of twitter.json x16; plus one core
    alone, plus the scalar port for scale.  The C calls release the GIL."""
    import threading
    import numpy as np
    from oracle import oracle
    oracle.build()
    reps = 16
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    base = np.frombuffer(doc * reps, dtype=np.uint8)
    avx = oracle.avx512_supported()

    def make_fn():
        if avx:
            out = np.empty(base.size + 128, dtype=np.uint32)
            return lambda a: oracle.stage1_avx512(a, out=out)
        return lambda a: oracle.stage1(a)

    def one_core(fn, secs):
        fn(base)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < secs:
            fn(base)
            n += base.size
        return n / (time.perf_counter() - t0) / 1e9

    one = one_core(make_fn(), 1.5)
    scalar_one = one_core(lambda a: oracle.stage1(a), 1.0) if avx else one
    samples = [base.copy() for _ in range(cores)]
    fns = [make_fn() for _ in range(cores)]
    done = [0] * cores
    stop = time.perf_counter() + seconds

    def work(k):
        while time.perf_counter() < stop:
            fns[k](samples[k])
            done[k] += samples[k].size

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    el = time.perf_counter() - t0
    total = sum(done)
    what = ("oracle/sj_avx512.c (AVX-512 restatement of the reference's 512-bit Java Vector-API stage 1: vpcmpb/kmov classes, "
            "vpshufb nibble tables, two passes like SimdJsonParser.stage1)" if avx else
            "oracle/sj_oracle.c (scalar C port; this host CPU has no AVX-512)")
    port = {"value": round(total / el / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "one_core": round(one, 3), "scalar_port_one_core": round(scalar_one, 3),
            "sample": "twitter.json x%d (%d B) per thread, %d threads, %d scans in %.1f s by %s; the reference itself needs a "
                      "JVM: %s" % (reps, base.size, cores, total // base.size, el, what, probe_java())}
    # the reference itself, the moment a JDK >= 24 and its sources are on the box (java/bench/org/simdjson/RefStage1Bench.java)
    ref, why = reference_jvm_baseline(doc)
    if ref is None:
        port["reference_jvm"] = "not run: " + why
        return port
    return {"value": round(ref["stage1_gb_per_s_all_threads"], 3), "unit": "GB/s", "cores": ref["cores"], "kind": "reference",
            "one_core": round(ref["stage1_gb_per_s_one_thread"], 3),
            "parse_per_s": round(ref["parse_per_s_all_threads"], 1), "parse_per_s_one_core": round(ref["parse_per_s_one_thread"], 1),
            "sample": "the reference's own Utf8Validator.validate + StructuralIndexer.index (SimdJsonParser.java:55-58) at %s-bit vectors on "
                      "JDK %s, one parser per thread over its own %d B slice of twitter.json copies, %.0f s per leg (%s)"
                      % (ref.get("vector_bits"), ref.get("java"), ref.get("slice_bytes", 0), ref.get("seconds", 0), ref.get("command")),
            "port": port}


# ---------------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------------
def roofline(alg_bytes, ms, input_bytes, kernel, launches, traffic=None, **more):
    """`achieved` follows SURVEY.md 8(d): input bytes (read once) per kernel second; the traffic form (all algorithmic
    bytes: input + outputs written once) is given beside it."""
    s = ms / 1e3
    r = {"bound": "hbm", "achieved": round(input_bytes / s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(input_bytes / s / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": kernel,
         "avg_kernel_ms": round(ms, 4), "launches": launches, "algorithmic_bytes_per_launch": int(alg_bytes),
         "achieved_all_algorithmic_bytes": round(alg_bytes / s / 1e9, 2),
         "frac_all_algorithmic_bytes": round(alg_bytes / s / 1e9 / HBM_PEAK_GBS, 4)}
    if traffic is not None:
        r["traffic_source"] = pmc_source()
        r["traffic_stale"] = pmc_stale()
    r.update(more)
    return r


def pmc_traffic(key):
    """HBM bytes per launch from the committed PMC passes of this round (rocprofv3 cannot run inside the timed region)."""
    try:
        with open(os.path.join(PROFILE_DIR, "pmc_summary.json")) as f:
            return int(json.load(f)[key]["hbm_traffic_bytes_per_launch"]["total"])
    except (OSError, KeyError, ValueError, TypeError):
        return None


def pmc_stale():
    """True when the kernels' sources have changed since the PMC summary was collected (tools/csrc_digest.py): `traffic` then
    describes other kernels than the ones this run timed."""
    import csrc_digest
    try:
        with open(os.path.join(PROFILE_DIR, "pmc_summary.json")) as f:
            return bool(csrc_digest.traffic_stale(json.load(f), ROOT)[0])
    except (OSError, ValueError):
        return True


def pmc_source():
    """where `traffic` comes from: NOT measured by this run -- the tracked rocprofv3 --pmc summary and the commit it was taken at"""
    try:
        with open(os.path.join(PROFILE_DIR, "pmc_summary.json")) as f:
            meta = json.load(f).get("_collected", {})
        return "profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_round.sh, calibrated: " \
               "pmc_calibration.json; collected at commit %s)" % (PROFILE_ROUND, meta.get("commit", "unknown"))
    except (OSError, ValueError):
        return None


# ---------------------------------------------------------------------------------------------------------------
# CPU legs of the extras: the reference's path restated in C (oracle/: AVX-512 stage 1 + the scalar stage 2 / StringParser),
# timed on this box's host cores on a bounded sample, one core and all cores
# ---------------------------------------------------------------------------------------------------------------
def _all_cores(make_work, seconds, cores):
    """`cores` threads, each looping its own work() (a C call that releases the GIL) until the deadline -> calls per second"""
    import threading
    works = [make_work() for _ in range(cores)]
    done = [0] * cores
    stop = time.perf_counter() + seconds

    def run(k):
        while time.perf_counter() < stop:
            works[k]()
            done[k] += 1

    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(k,)) for k in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return sum(done) / (time.perf_counter() - t0)


def _one_core(work, seconds):
    work()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        work()
        n += 1
    return n / (time.perf_counter() - t0)


def cpu_legs(doc, pool_unit, pool_offs):
    """-> {section: cpu_baseline object}.  parse = SimdJsonParser.parse of twitter.json (ParseBenchmark.java:40-48's shape);
    strings = every StringParser.parseString of twitter.json; batch = parse of ~1 KB documents."""
    import numpy as np
    from oracle import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    avx = oracle.avx512_supported()
    kind = "AVX-512 stage 1 (oracle/sj_avx512.c) + scalar stage 2 (oracle/sj_oracle.c)" if avx else "scalar port (oracle/sj_oracle.c)"
    one = np.array([0, len(doc)], dtype=np.uint64)
    darr = np.frombuffer(doc + b"\0" * 64, dtype=np.uint8)
    parse1 = _one_core(lambda: oracle.parse_many(darr, one, 1, avx), 1.5)
    parse_all = _all_cores(lambda: (lambda a=darr.copy(): oracle.parse_many(a, one, 1, avx)), 3.0, cores)
    idx0, _ = oracle.stage1(doc)
    sb = np.zeros(len(doc) + 4 * idx0.size + 128, dtype=np.uint8)
    str1 = _one_core(lambda: oracle.unescape_loop(darr, idx0, 4, sb), 1.0) * 4
    def mk_str():
        a, s2 = darr.copy(), sb.copy()
        return lambda: oracle.unescape_loop(a, idx0, 4, s2)
    str_all = _all_cores(mk_str, 2.0, cores) * 4
    parr = np.frombuffer(bytes(pool_unit) + b"\0" * 64, dtype=np.uint8)
    nd = pool_offs.size - 1
    batch1 = _one_core(lambda: oracle.parse_many(parr, pool_offs, 1, avx), 1.5) * nd
    batch_all = _all_cores(lambda: (lambda a=parr.copy(): oracle.parse_many(a, pool_offs, 1, avx)), 3.0, cores) * nd
    note = "kind 'port': %s; the reference itself needs a JVM: %s" % (kind, probe_java())
    parse = {"value": round(parse_all, 1), "unit": "parses/s", "cores": cores, "kind": "port", "one_core": round(parse1, 1),
             "one_core_ms_per_parse": round(1e3 / parse1, 4),
             "sample": "SimdJsonParser.parse(twitter.json) = stage 1 + stage 2 (JsonIterator + TapeBuilder + StringParser), %d threads x 3 s; %s" % (cores, note)}
    strings = {"value": round(str_all * len(doc) / 1e9, 3), "unit": "GB/s of document", "cores": cores, "kind": "port",
               "one_core": round(str1 * len(doc) / 1e9, 3),
               "sample": "StringParser.parseString for the 18,099 strings of twitter.json over given indexes, %d threads x 2 s; %s" % (cores, note)}
    batch = {"value": round(batch_all, 1), "unit": "docs/s", "cores": cores, "kind": "port", "one_core": round(batch1, 1),
             "sample": "stage 1 + stage 2 of %d unique ~1 KB documents (the configs[3] pool), one after the other per thread, %d threads x 3 s; %s" % (nd, cores, note)}
    return {"parse": parse, "strings": strings, "batch": batch}



class Stage1Runner:
    """k_stage1 over one resident buffer through the C ABI, timed with HIP events attached to the dispatch."""

    def __init__(self, torch, S, dev, work, buf, n, s_total):
        self.torch, self.work, self.buf, self.n = torch, work, buf, n
        self.cap = s_total + 1
        self.out = torch.empty(self.cap, dtype=torch.int32, device=dev)
        self.res = torch.zeros(2, dtype=torch.int64, device=dev)
        self.ctx = S.Context(device=dev.index, capacity=1 << 20)

    def launch(self, k=1):
        for _ in range(k):
            self.ctx.stage1_device(self.buf.data_ptr(), self.n, self.out.data_ptr(), self.cap, self.res.data_ptr(),
                                   self.work.cuda_stream)

    def kernel_ms(self, k):
        """average kernel time of k launches (HIP events on the launch stream)"""
        self.ctx.set_profiling(True)
        self.launch(k)
        self.torch.cuda.synchronize()
        ms, launches = self.ctx.kernel_time()
        self.ctx.set_profiling(False)
        return ms / max(launches, 1)

    def cold_and_settled(self, steps, settle_ms=60.0):
        """cold: the first `steps` launches after the GPU idled for 0.6 s; settled: `steps` launches after at least
        settle_ms of back-to-back launches (the clock governor needs ~25 ms of this kernel, tools/perlaunch.py)."""
        self.torch.cuda.synchronize()
        time.sleep(0.6)
        cold = self.kernel_ms(steps)
        self.launch(max(steps, int(settle_ms / max(cold, 1e-3)) + 1))
        settled = self.kernel_ms(steps)
        return cold, settled

    def status(self):
        r = self.res.cpu().numpy()
        return int(r[0]), int(r[1]) & 0xFFFFFFFF


def wall_steps(torch, fn, steps, dist=None):
    """K steps bracketed by barrier + synchronize on both sides -> seconds (max over ranks)."""
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el


BATCH_GEN_SLICE = 100000  # documents generated (and page-locked) at a time: ~130 MB of pinned host memory per rank, whatever --docs is


def make_batch_shard(torch, S, W, dev, ctx, lo, hi):
    """documents [lo, hi) of the configs[3] set (1,000,000 UNIQUE documents, tools/docgen.c seed 20250825: document k is a
    function of (seed, k), so a rank generates its own range only) as a BatchShard on `dev` -> (shard, local offsets).
    Generated in slices of BATCH_GEN_SLICE documents through ONE re-used page-locked buffer: eight ranks of a node with a million
    documents each would otherwise pin 8 x 1.3 GB of host memory at the same time."""
    import numpy as np
    from simdjson_java_amd import sharding
    lens = W.unique_doc_lengths(lo, hi - lo)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(offs[-1])
    shard_bytes = torch.empty(n, dtype=torch.uint8, device=dev)
    host = torch.empty(1300 * int(os.environ.get("SJMI_DOC_SCALE", "1")) * min(BATCH_GEN_SLICE, hi - lo) + 64, dtype=torch.uint8).pin_memory()
    for a in range(0, hi - lo, BATCH_GEN_SLICE):
        m = min(BATCH_GEN_SLICE, hi - lo - a)
        data, o2 = W.unique_docs(lo + a, m, out=host.numpy())
        b0, b1 = int(offs[a]), int(offs[a + m])
        assert int(o2[-1]) == b1 - b0
        shard_bytes[b0:b1].copy_(host[:b1 - b0], non_blocking=True)
        torch.cuda.synchronize()  # (the buffer is re-used by the next slice)
    del host
    # (this set: 5.4 B per structural, 0.84 string-buffer bytes and 0.13 tape words per input byte)
    return sharding.BatchShard(ctx, shard_bytes, offs, dev, index_ratio=4, string_ratio=1.0, tape_ratio=0.2), offs


def check_batch_sample(torch, oracle, shard, offs, sample=10000, seed=20250825):
    """per-document parity of a seeded sample of the batch against the oracle, outside every timed region: structural
    indexes (document-relative), tape / string records as a tree (oracle.Parsed.to_python: type, int64, raw double bits,
    UTF-8 bytes, order), the document's error code == 0"""
    import random
    import numpy as np
    n_docs = shard.n_docs
    ks = sorted(random.Random(seed).sample(range(n_docs), min(sample, n_docs)))
    io = shard.index_offsets.cpu().numpy()
    to = shard.tape_offsets.cpu().numpy()
    err = shard.doc_errors.cpu().numpy()
    c = shard.check()
    sb = bytes(shard.sb[:c["string_bytes"]].cpu().numpy())
    host_buf = shard.buf[:shard.n].cpu().numpy()
    idx = shard.idx[:c["structurals"]].cpu().numpy().view(np.uint32)
    tape = shard.tape[:int(to[-1])].cpu().numpy().view(np.uint64)
    for k in ks:
        a, b = int(offs[k]), int(offs[k + 1])
        d = host_buf[a:b].tobytes()
        w_idx, w_st = oracle.stage1(d)
        assert w_st == 0 and int(err[k]) == 0, k
        got = idx[int(io[k]):int(io[k + 1])].astype(np.int64) - a
        assert np.array_equal(got, w_idx.astype(np.int64)), "document %d: indexes differ from the oracle's" % k
        want = oracle.parse(d)
        assert want.error == 0
        assert oracle.Parsed(tape[int(to[k]):int(to[k + 1])], sb, 0, 0, 0).to_python() == want.to_python(), \
            "document %d: tree differs from the oracle's" % k
    return len(ks)


def check_batch_all(torch, oracle, shard, offs):
    """EVERY document of the batch against the oracle, outside every timed region: per document a digest of its tape (every word;
    STRING words by the record's bytes instead of the buffer offset: oracle/sj_oracle.c sjo_tape_digest), a hash of its structural
    indexes relative to its first byte, the structural count and the error code -- the oracle's from its own per-document parse
    (sjo_digest_many, on the host's cores), the engine's from the batch's output arrays (sjo_digest_outputs)."""
    import numpy as np
    c = shard.check()
    n_docs = shard.n_docs
    host_buf = shard.buf[:shard.n].cpu().numpy()
    want_dig, want_err, want_ih, want_cnt = oracle.digest_many(host_buf, np.asarray(offs, dtype=np.uint64))
    to = shard.tape_offsets.cpu().numpy().view(np.uint64)
    io = shard.index_offsets.cpu().numpy().view(np.uint64)
    err = shard.doc_errors.cpu().numpy()[:n_docs]
    tape = shard.tape[:int(to[-1])].cpu().numpy().view(np.uint64)
    sb = shard.sb[:c["string_bytes"]].cpu().numpy()
    idx = shard.idx[:c["structurals"]].cpu().numpy().view(np.uint32)
    got_dig, got_ih, got_cnt = oracle.digest_outputs(tape, to, sb, idx, io, np.asarray(offs, dtype=np.uint64), err)
    assert np.array_equal(err, want_err), "document verdicts differ from the oracle's at %s" % np.flatnonzero(err != want_err)[:5]
    # (a document that fails stage 1 -- codes 1..3 -- has NO structurals in the engine's isolated mode, include/sjmi.h; the
    #  reference throws before anyone sees its indexes, SimdJsonParser.java:55-58)
    s1 = (want_err >= 1) & (want_err <= 3)
    assert not got_cnt[s1].any(), "a document that failed stage 1 has structurals"
    assert np.array_equal(got_cnt[~s1], want_cnt[~s1]) and np.array_equal(got_ih[~s1], want_ih[~s1]), \
        "structural indexes differ from the oracle's at documents %s" % np.flatnonzero(~s1 & ((got_cnt != want_cnt) | (got_ih != want_ih)))[:5]
    assert np.array_equal(got_dig, want_dig), "tapes / string records differ from the oracle's at documents %s" % np.flatnonzero(got_dig != want_dig)[:5]
    return n_docs


BATCH_KERNELS = ("k_stage1_batch (one plain pass with per-block side outputs, accepted on the device) + k_strings<true> + k_doc_prepare "
                 "(separator check, index ranges, string ordinals, predicted tape lengths) + tape-offset scan + k_tok_stream (token walker over runs of documents; "
                 "k_coop_walk in list mode behind it for the documents it declines)")


def batch_algorithmic_bytes(n, c):
    """input read once + uint32 indexes + string records + tape words, each written once"""
    return n + 4 * c["structurals"] + c["string_bytes"] + 8 * c["tape_words"]


# ---------------------------------------------------------------------------------------------------------------
# N = 1
# ---------------------------------------------------------------------------------------------------------------
def bench_single(args):
    import numpy as np
    import torch
    import simdjson_java_amd as S
    import workloads as W
    from oracle import oracle  # checker only (closed forms of ONE tile) + the cpu_baseline leg

    oracle.build()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    work = torch.cuda.Stream(device=dev)
    work.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(work)  # one explicit stream for everything: kernels, torch ops, events

    doc = W.load_twitter()
    n0 = len(doc)
    idx0, st0 = oracle.stage1(doc)
    assert st0 == 0 and idx0.size == 55263
    reps = args.reps
    buf, n = W.repeat_on_device(doc, reps, dev)
    assert n < (1 << 32)
    s_total = idx0.size * reps
    r1 = Stage1Runner(torch, S, dev, work, buf, n, s_total)
    assert r1.ctx.selftest() == 0
    if args.tile_steps:
        r1.ctx.set_tile_steps(args.tile_steps)
    # ---- parity, outside every timed region: the whole uint32 index array against the closed form ----
    r1.launch()
    torch.cuda.synchronize()
    cnt, st = r1.status()
    assert cnt == s_total and st == 0, (cnt, st)
    ok, bad = W.closed_form_ok(r1.out, idx0, n0, reps)
    assert ok, "GPU indexes differ from the oracle's closed form (copies %d..)" % bad
    assert int(r1.out[s_total].item()) == 0
    sections = set() if args.no_extras else set(x for x in args.sections.split(",") if x)
    # ---- the dominant kernel, cold and settled ----
    cold_ms, settled_ms = (1.0, 1.0) if args.skip_main_timing else r1.cold_and_settled(args.steps)
    # ---- the contract's timed region: W warmup steps, then exactly K steps ----
    r1.launch(0 if args.skip_main_timing else args.warmup + args.preheat)
    r1.ctx.set_profiling(True)
    elapsed = wall_steps(torch, r1.launch, args.steps)
    kern_ms, launches = r1.ctx.kernel_time()
    r1.ctx.set_profiling(False)
    timed_kernel_ms = kern_ms / max(launches, 1)
    cnt, st = r1.status()
    assert cnt == s_total and st == 0
    alg = n + 4 * (s_total + 1)
    name = "twitter.json x%d" % reps
    line = {
        "metric": "GB/s JSON scanned (stage-1)", "value": round(n * args.steps / elapsed / 1e9, 2), "unit": "GB/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic: twitter.json (631,515 B reference fixture) x%d byte-concatenated (%d B), resident in HBM" % (reps, n),
        "config": {"workload": "%s concatenated (north-star '4 GiB concatenated twitter.json'): stage-1 = UTF-8 validation + "
                               "structural indexing + uint32 index compaction, one fused single-pass kernel; the full index "
                               "array is checked against the oracle's closed form on the device before timing" % name,
                   "bytes_per_gpu": n, "structurals_per_gpu": s_total, "tile_steps": args.tile_steps or "auto",
                   "preheat_launches": args.preheat, "sharding": "none"},
        "roofline": roofline(alg, timed_kernel_ms, n, "k_stage1", launches, traffic=pmc_traffic("stage1_twitter_4g") if reps == W.TWITTER_4G_REPS else None,
                             timed_region="the K timed steps (after W warmup + preheat_launches untimed steps)",
                             settled={"avg_kernel_ms": round(settled_ms, 4), "frac": round(n / settled_ms / 1e6 / HBM_PEAK_GBS, 4),
                                      "protocol": "K launches after >= 60 ms of back-to-back launches"},
                             cold={"avg_kernel_ms": round(cold_ms, 4), "frac": round(n / cold_ms / 1e6 / HBM_PEAK_GBS, 4),
                                   "protocol": "the first K launches after the GPU idled 0.6 s, no warmup at all"}),
    }
    extra = {}
    if sections & {"x1024", "unescape"}:
        # ---- configs[1]: twitter x1024 = the first 1024 copies of the same buffer (bytes >= len are invisible) ----
        reps1 = min(1024, reps)
        n1, s1 = n0 * reps1, idx0.size * reps1
        ra = Stage1Runner(torch, S, dev, work, buf, n1, s1)
        ra.launch()
        torch.cuda.synchronize()
        assert ra.status() == (s1, 0)
        ok, bad = W.closed_form_ok(ra.out, idx0, n0, reps1)
        assert ok
        c1, s1ms = ra.cold_and_settled(max(args.steps, 50)) if "x1024" in sections else (1.0, 1.0)
        if "x1024" in sections:
            extra["stage1_twitter_x1024"] = {
                "config": "configs[1]: twitter.json x%d (%d B), stage 1, bit-exact index check" % (reps1, n1),
                "value": round(n1 / s1ms / 1e6, 2), "unit": "GB/s",
                "roofline": roofline(n1 + 4 * (s1 + 1), s1ms, n1, "k_stage1", max(args.steps, 50), traffic=pmc_traffic("stage1_twitter_x1024"),
                                     cold={"avg_kernel_ms": round(c1, 4), "frac": round(n1 / c1 / 1e6 / HBM_PEAK_GBS, 4)})}
    if "unescape" in sections:
        # ---- string unescape (StringParser.parseString for every string) on twitter x1024 ----
        _, _, masks = oracle.index_blocks(doc, want_masks=True)
        n_str0 = int((np.frombuffer(doc, dtype=np.uint8)[idx0] == 0x22).sum())
        raw0 = int(sum(bin(int(x)).count("1") for x in masks[:, 2])) + n_str0  # bytes inside quotes + both quotes
        want_sb, _, feo, _ = oracle.unescape_all(doc + b"\0" * 64, idx0)
        assert feo < 0
        sb_cap = n1 + 4 * (s1 + 1) + 64
        sb = torch.empty(sb_cap, dtype=torch.uint8, device=dev)
        ures = torch.zeros(3, dtype=torch.int64, device=dev)

        def unescape():
            ra.ctx.unescape_device(buf.data_ptr(), n1, ra.out.data_ptr(), s1, sb.data_ptr(), sb_cap, ures.data_ptr(), work.cuda_stream)

        unescape()
        torch.cuda.synchronize()
        u = ures.cpu().numpy()
        assert int(u[0]) == len(want_sb) * reps1 and int(u[1]) == 0, u
        sbv = sb[:len(want_sb) * reps1].view(reps1, len(want_sb))
        assert torch.equal(sbv[0].cpu(), torch.frombuffer(bytearray(want_sb), dtype=torch.uint8)), "string buffer differs from the oracle's"
        assert torch.equal(sbv, sbv[:1].expand(reps1, len(want_sb)))
        for _ in range(20):
            unescape()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            unescape()
        e1.record()
        torch.cuda.synchronize()
        ums = e0.elapsed_time(e1) / 20
        ualg = reps1 * (raw0 + 4 * n_str0 + len(want_sb))
        extra["unescape_twitter_x1024"] = {
            "config": "StringParser.parseString for all %d strings of twitter.json x%d: [be32 length][unescaped bytes] records, "
                      "byte-identical to the oracle's string buffer" % (n_str0 * reps1, reps1),
            "value": round(n1 / ums / 1e6, 2), "unit": "GB/s of document",
            "roofline": {"bound": "hbm", "achieved": round(ualg / ums / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ualg / ums / 1e6 / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("unescape_twitter_x1024"),
                         "traffic_source": pmc_source(), "traffic_stale": pmc_stale(),
                         "kernel": "k_strings (one streaming pass; + the memsets of its chain state and result record)", "avg_ms_per_call": round(ums, 4),
                         "algorithmic_bytes_per_launch": ualg,
                         "algorithmic_bytes": "string bytes incl. quotes read (%d) + 4 B index word per string + records written (%d), per copy"
                                              % (raw0, len(want_sb))}}
        del sb
    if "synth" in sections:
        # ---- configs[2]: 4 GiB synthetic ----
        tile = W.synth_tile()
        tidx, tst, tmasks = oracle.index_blocks(tile, want_masks=True)
        assert tst == 0
        tfrac = W.measured_fractions(tile, tmasks)
        treps = ((1 << 32) - 1) // len(tile)
        tbuf, tn = W.repeat_on_device(tile, treps, dev)
        rs = Stage1Runner(torch, S, dev, work, tbuf, tn, tidx.size * treps)
        rs.launch()
        torch.cuda.synchronize()
        assert rs.status() == (tidx.size * treps, 0)
        ok, bad = W.closed_form_ok(rs.out, tidx, len(tile), treps)
        assert ok
        c2, s2 = rs.cold_and_settled(args.steps)
        extra["stage1_synthetic_4g"] = {
            "config": "configs[2]: 4 GiB synthetic JSON per SURVEY.md 8(d) (tools/synth.synth_tile_spec seed 20250824: one %d B tile x%d = "
                      "%d B), stage 1 + UTF-8 validation, closed-form index check; measured on the tile: %s, %.2f B per structural"
                      % (len(tile), treps, tn, json.dumps(tfrac), len(tile) / tidx.size),
            "measured_fractions": tfrac,
            "value": round(tn / s2 / 1e6, 2), "unit": "GB/s",
            "roofline": roofline(tn + 4 * (tidx.size * treps + 1), s2, tn, "k_stage1", args.steps, traffic=pmc_traffic("stage1_synthetic_4g"),
                                 cold={"avg_kernel_ms": round(c2, 4), "frac": round(tn / c2 / 1e6 / HBM_PEAK_GBS, 4)})}
        del tbuf, rs
    if "batch" in sections:
        # ---- configs[3] on ONE GPU: the batched path, 1,000,000 documents ----
        extra["batch_1m_docs"] = batch_single_gpu(torch, S, W, dev, work, args, with_h2d=not args.batch_accepted_only,
                                                  rejected=not args.batch_accepted_only)
    if "parse" in sections:
        # ---- the drop-in call itself: SimdJsonParser.parse(byte[], len) of ONE twitter.json from a host buffer (H2D, all
        #      stages, outputs back on the host), with the reference's stage 2 on the host or all three stages on the GPU ----
        try:  # (needs g++ for the user-side C++ of tools/ondemand_bench.cpp: an extra must not cost the line)
            extra["parse_single_document"] = parse_single_document(S, doc)
        except (OSError, subprocess.CalledProcessError) as e:
            extra["parse_single_document"] = {"error": "not measured: %s" % e}
    if "select" in sections:
        # ---- the reference's headline benchmark shape (jmh ParseAndSelectBenchmark / SchemaBasedParseAndSelectBenchmark: the
        #      screen names of twitter.json's users with default_profile), user code in C++ against the public C ABI ----
        import ondemand_bench
        try:  # (needs g++ when tools/libondemand_bench.so did not travel with the tree: an extra must not cost the line)
            sel = ondemand_bench.measure(doc)
            assert all(v["selected"] == 86 for v in sel.values()), sel  # BenchmarkCorrectnessTest.java:23-55
            extra["parse_and_select_twitter_json"] = {
                "config": "twitter.json from a host buffer -> the 86 screen names of users with default_profile, per call: H2D, GPU "
                          "stage 1 (+ string records and host stage 2 for the full parse), selection on the host through sjmi_value_* / "
                          "sjmi_od_* (tools/ondemand_bench.cpp)",
                "unit": "ms per parse-and-select", "value": sel["on_demand_scan"]["ms"], **sel,
                "reference_readme": "README.md, 512-bit vectors, Xeon Platinum 8375C, one thread: SchemaBasedParseAndSelectBenchmark "
                                    "3164 ops/s, ParseAndSelectBenchmark 1842 ops/s (other hardware; no JVM here)"}
        except (OSError, subprocess.CalledProcessError) as e:
            extra["parse_and_select_twitter_json"] = {"error": "not measured: %s" % e}
    if "trees" in sections:
        # ---- configs[4]: twitter x1024 as 1024 documents from a HOST buffer -> 1024 trees (GPU stage 1 + GPU string records +
        #      the host stage 2 on a thread pool), timed from C++; every tree then selected on through sjmi_value_* ----
        try:
            extra["twitter_x1024_as_1024_trees"] = trees_1024(S, doc)
        except (OSError, subprocess.CalledProcessError) as e:
            extra["twitter_x1024_as_1024_trees"] = {"error": "not measured: %s" % e}
    if not args.no_cpu_baseline:
        # ---- the CPU beside every section (bounded samples, about 25 s in all) ----
        pdocs, punit, plens = W.small_doc_pool(args.pool)
        legs = cpu_legs(doc, punit, W.batch_offsets(plens, 1))
        for key, leg in (("unescape_twitter_x1024", "strings"), ("batch_1m_docs", "batch"), ("parse_single_document", "parse"),
                         ("parse_and_select_twitter_json", "parse"), ("twitter_x1024_as_1024_trees", "parse")):
            if key in extra:
                extra[key]["cpu_baseline"] = legs[leg]
        # configs[0]: the reference's own CPU-runnable case (plumbing, no GPU): twitter.json through the CPU restatement,
        # checked against the survey's counts (S = 55,263 structurals, 86 users with default_profile)
        want = oracle.parse(doc)
        _, unique_users = _twitter_default_profile_users(want.to_python())
        assert want.error == 0 and idx0.size == 55263 and unique_users == 86  # BenchmarkCorrectnessTest.java:19-42
        extra["configs0_cpu_parse_twitter_json"] = {
            "config": "configs[0]: twitter.json (631,515 B) single-document parse on the CPU path (no GPU): 55,263 structurals, 86 unique "
                      "users with default_profile", "value": legs["parse"]["one_core_ms_per_parse"], "unit": "ms per parse (one core)",
            "cpu_baseline": legs["parse"]}
    line["extra"] = extra
    # the other configs' figures once more as flat scalars (a driver that keeps only the contract's keys keeps `config` and
    # `roofline` but not `extra` or nested objects)
    rf = line["roofline"]
    if reps == W.TWITTER_4G_REPS and not args.no_extras:
        # the memory system's rate for THIS traffic on THIS box, beside the nominal peak: a kernel that does nothing but read 4 GiB in
        # 4 KiB steps per wave and store 0.34 bytes per byte read in 5.6 KB bursts (tools/ubench/load_pattern.hip, built by
        # __graft_entry__.build(); a reported yardstick, never part of `value` or `frac`)
        ms = memory_system_rate()
        if ms:
            rf["same_traffic_do_nothing_kernel"] = ms
            rf["same_traffic_do_nothing_kernel_GBps_of_input"] = round(1e3 * max(ms["read_plus_index_stores_TBps_of_input"], ms["read_plus_streaming_index_stores_TBps_of_input"]), 1)
            rf["achieved_over_same_traffic_do_nothing_kernel"] = round(rf["achieved"] / rf["same_traffic_do_nothing_kernel_GBps_of_input"], 4)
    rf["settled_frac"], rf["settled_avg_kernel_ms"] = rf["settled"]["frac"], rf["settled"]["avg_kernel_ms"]
    rf["cold_frac"], rf["cold_avg_kernel_ms"] = rf["cold"]["frac"], rf["cold"]["avg_kernel_ms"]
    flat = line["config"]

    def put(key, sec, *path):
        v = extra.get(sec)
        for k in path:
            v = v.get(k) if isinstance(v, dict) else None
        if v is not None:
            flat[key] = v

    put("configs1_x1024_frac_settled", "stage1_twitter_x1024", "roofline", "frac")
    put("configs1_x1024_frac_cold", "stage1_twitter_x1024", "roofline", "cold", "frac")
    put("configs2_synthetic_4g_frac_settled", "stage1_synthetic_4g", "roofline", "frac")
    put("unescape_x1024_ms", "unescape_twitter_x1024", "roofline", "avg_ms_per_call")
    put("unescape_x1024_frac", "unescape_twitter_x1024", "roofline", "frac")
    put("configs3_batch_docs_per_s", "batch_1m_docs", "value")
    put("configs3_batch_ms", "batch_1m_docs", "ms_per_batch")
    put("configs3_batch_frac", "batch_1m_docs", "roofline", "frac")
    put("configs3_batch_documents", "batch_1m_docs", "documents")
    put("configs3_batch_oracle_checked_documents", "batch_1m_docs", "oracle_checked_documents")
    put("configs3_batch_oracle_digest_checked_documents", "batch_1m_docs", "oracle_digest_checked_documents")
    put("configs3_batch_docs_per_s_incl_h2d", "batch_1m_docs", "incl_h2d", "value")
    put("configs3_batch_incl_h2d_ms", "batch_1m_docs", "incl_h2d", "ms_per_batch")
    put("configs3_batch_incl_h2d_pcie_floor_ms", "batch_1m_docs", "incl_h2d", "pcie_floor_ms")
    put("configs3_batch_exact_ms", "batch_1m_docs", "rejected_path", "exact_ms")
    put("configs3_batch_one_bad_doc_ms", "batch_1m_docs", "rejected_path", "one_bad_doc_ms")
    put("configs3_batch_one_bad_doc_stage2_ms", "batch_1m_docs", "rejected_path", "one_bad_doc_stage2_ms")
    put("configs3_batch_no_separator_ms", "batch_1m_docs", "rejected_path", "no_separator_ms")
    put("configs3_batch_no_separator_latched_ms", "batch_1m_docs", "rejected_path", "no_separator_latched_ms")
    put("configs3_batch_one_bad_doc_over_accepted", "batch_1m_docs", "rejected_path", "one_bad_doc_over_accepted")
    put("configs3_batch_accepted_checked_each_step_ms", "batch_1m_docs", "rejected_path", "accepted_checked_each_step_ms")
    put("parse_twitter_json_all_device_ms", "parse_single_document", "twitter_json", "gpu_walker", "ms")
    put("parse_twitter_json_host_walker_ms", "parse_single_document", "twitter_json", "host_walker", "ms")
    put("configs4_1024_trees_ms", "twitter_x1024_as_1024_trees", "value")
    put("configs4_1024_trees_pcie_floor_ms", "twitter_x1024_as_1024_trees", "pcie_floor_ms")
    put("select_on_demand_ms", "parse_and_select_twitter_json", "value")
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(doc)
    print(json.dumps(line))
    r1.ctx.close()


def parse_single_document(S, doc, reps=300):
    """SimdJsonParser.parse of ONE document from a host buffer, both placements of stage 2; the call is timed from C++
    (tools/ondemand_bench.cpp mode 3: sjmi_parser_parse without the Python binding's copies), the tape is checked against the
    oracle through the binding once per mode.  twitter.json and a 16 MiB array of objects."""
    import ctypes as C
    import ondemand_bench
    from oracle import oracle as O
    L = ondemand_bench.load_bench_lib()
    big = b"[" + b",".join(b'{"id":%d,"name":"user %d","tags":["a","b"],"score":%d.5,"ok":true}' % (i, i, i % 97) for i in range(225000)) + b"]"
    out = {"config": "sjmi_parser_parse (= SimdJsonParser.parse(byte[], len)) from a host buffer, end to end per call: H2D of the document, "
                     "stage 1, string records, stage 2, tape + string buffer on the host; timed from C++, tape checked against the oracle",
           "unit": "ms per document"}
    for name, d, n in (("twitter_json", doc, reps), ("array_16mib", big, 20)):
        want = O.parse(d)
        buf = (C.c_uint8 * len(d)).from_buffer_copy(d)
        res = {"bytes": len(d)}
        for mode, key in ((False, "host_walker"), (True, "gpu_walker")):
            p = S.SimdJsonParser(capacity=len(d) + 64, gpu_walk=mode)
            tape = p.parse(d).tape
            if tape.size != want.tape.size or not (tape == want.tape).all():
                raise SystemExit("parse(%s) tape differs from the oracle's (gpu_walk=%s)" % (name, mode))
            secs, a, b = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
            ms = 1e9
            for rep in range(4):  # (first round untimed: clocks)
                if L.odb_run(p._h, buf, len(d), 3, n, C.byref(secs), C.byref(a), C.byref(b)):
                    raise SystemExit("sjmi_parser_parse failed in the parse section")
                if rep:
                    ms = min(ms, secs.value / n * 1e3)
            res[key] = {"ms": round(ms, 4), "GB/s": round(len(d) / ms / 1e6, 3)}
            p.close()
        out[name] = res
    out["value"] = out["twitter_json"]["gpu_walker"]["ms"]
    return out


def _twitter_default_profile_users(tree):
    """the oracle's tree (tagged tuples: ("o", size, [(key, value)...]), ("a", size, [...]), ("s", bytes), ("t",) ...) ->
    (statuses whose user has default_profile, their unique screen names): BenchmarkCorrectnessTest.java:23-42"""
    def field(obj, name):
        assert obj[0] == "o"
        for k, v in obj[2]:
            if k == name or k == name.encode():
                return v
        raise KeyError(name)
    names = []
    for st in field(tree, "statuses")[2]:
        user = field(st, "user")
        if field(user, "default_profile")[0] == "t":
            names.append(field(user, "screen_name")[1])
    return len(names), len(set(names))


def trees_1024(S, doc, reps=1024, iters=3):
    """BASELINE.json configs[4]"""
    import ctypes as C
    import numpy as np
    import ondemand_bench
    from oracle import oracle as O
    L = ondemand_bench.load_bench_lib()
    d1 = doc.rstrip() + b"\n"
    host = np.frombuffer(d1 * reps, dtype=np.uint8)
    offs = np.arange(reps + 1, dtype=np.uint64) * np.uint64(len(d1))
    want_users, _ = _twitter_default_profile_users(O.parse(doc).to_python())
    p = S.SimdJsonParser(capacity=host.size + 64)
    secs, ok, umin, umax = C.c_double(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    best = 1e9
    for rep in range(3):  # (first round: allocations, page-locking, clocks)
        rc = L.odb_parse_batch(p._h, host.ctypes.data, host.size, offs.ctypes.data, reps, 1 if rep == 0 else iters, C.byref(secs),
                               C.byref(ok), C.byref(umin), C.byref(umax))
        if rc:
            raise SystemExit("sjmi_parser_parse_batch failed in the trees section: %d" % rc)
        if rep:
            best = min(best, secs.value / iters * 1e3)
    assert ok.value == reps and umin.value == umax.value == want_users, (ok.value, umin.value, umax.value, want_users)
    p.close()
    return {"config": "configs[4]: twitter.json x%d as %d documents (%d B) from a host buffer -> %d trees: H2D, GPU stage 1 (per-document "
                      "verdicts) + GPU string records, D2H, host stage 2 (JsonIterator / TapeBuilder mirror) on a thread pool, pipelined "
                      "over sub-batches (sjmi_parser_parse_batch, timed from C++); every tree walked through sjmi_value_*: %d statuses "
                      "with default_profile each (the oracle's count); word-for-word tree equality: tests/test_gpu_fullscale.py"
                      % (reps, reps, host.size, reps, want_users),
            "value": round(best, 3), "unit": "ms per batch", "docs_per_s": round(reps / best * 1e3, 1), "GB_per_s": round(host.size / best / 1e6, 3),
            # what the link alone costs: the batch in, indexes (4 B per structural) + string records out, one direction at a time
            # at the ~57 GB/s tools/pcie.py measures on these boxes (both directions at once halve each): NOT GPU time
            "pcie_floor_ms": round((host.size + 4 * 55263 * reps + 440313 * reps) / 57e9 * 1e3, 2),
            "pcie_floor": "bytes in + uint32 indexes + string records out = %d MB at 57 GB/s one way at a time (tools/pcie.py; the "
                          "link's rate differs by +-5 %% between boxes, so a batch can come in just under this figure); the GPU kernels "
                          "of this batch take ~1 ms" % ((host.size + 4 * 55263 * reps + 440313 * reps) // 1000000)}


def batch_single_gpu(torch, S, W, dev, work, args, with_h2d=True, n_docs=None, check=True, rejected=True):
    from oracle import oracle  # (checker of the sample, outside the timed region)
    n_docs = n_docs or args.docs
    ctx = S.Context(device=dev.index, capacity=1 << 20)
    shard, offs = make_batch_shard(torch, S, W, dev, ctx, 0, n_docs)
    st = work.cuda_stream
    for _ in range(3):
        shard.step(st)
    torch.cuda.synchronize()
    c = shard.check()
    assert c["failed_documents"] == 0 and c["host_documents"] == 0 and c["stage1_status"] == 0, c
    checked = check_batch_sample(torch, oracle, shard, offs, args.sample) if check else 0
    checked_all = check_batch_all(torch, oracle, shard, offs) if check else 0
    # (like the headline's preheat_launches: the clock governor needs ~25 ms of work before the figure is a steady-state one --
    #  untimed, disclosed in the section as `preheat_steps`; the first timed step of a cold GPU is ~8 % slower)
    preheat = getattr(args, "batch_preheat", 0)
    for _ in range(preheat):
        shard.step(st)
    el = wall_steps(torch, lambda: shard.step(st), args.batch_steps)
    ms = el / args.batch_steps * 1e3
    alg = batch_algorithmic_bytes(shard.n, c)
    out = {"config": "configs[3] on one GPU: %d UNIQUE documents (tools/docgen.c, seed 20250825, lengths uniform in [768, 1280] B, %d B, "
                     "%.1f structurals per document), device-resident: stage 1 (per-document verdicts) -> string records -> GPU walk "
                     "(tapes), sjmi_parse_batch_device_optimistic; before timing ALL %d documents compared per document with the oracle (tape digest incl. "
                     "string bytes, index hash, verdict) and %d seeded sample documents as trees"
                     % (n_docs, shard.n, c["structurals"] / n_docs, checked_all, checked),
           "value": round(n_docs / (ms / 1e3), 1), "unit": "docs/s", "ms_per_batch": round(ms, 3), "counts": c,
           "documents": n_docs, "oracle_checked_documents": checked, "oracle_digest_checked_documents": checked_all,
           "timed_steps": args.batch_steps, "preheat_steps": preheat,
           "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("batch_1m_docs"),
                        "traffic_source": pmc_source(), "traffic_stale": pmc_stale(),
                        "kernel": BATCH_KERNELS,
                        "algorithmic_bytes_per_launch": alg,
                        "algorithmic_bytes": "input read once + uint32 indexes + string records + tape words written once"}}
    if check and rejected:
        out["rejected_path"] = batch_rejected_path(torch, oracle, shard, offs, st, ms, max(3, args.batch_steps // 4))
    if with_h2d:
        host = torch.empty(shard.n, dtype=torch.uint8).pin_memory()
        host.copy_(shard.buf[:shard.n])

        def step_h2d():
            shard.buf[:shard.n].copy_(host, non_blocking=True)
            shard.step(st)

        k2 = max(3, args.batch_steps // 2)
        step_h2d()
        el = wall_steps(torch, step_h2d, k2)
        ms2 = el / k2 * 1e3
        # the link's floor: the same copy alone
        el = wall_steps(torch, lambda: shard.buf[:shard.n].copy_(host, non_blocking=True), k2)
        floor_ms = el / k2 * 1e3
        # ... and the copy of batch k + 1 UNDER the kernels of batch k: two device input buffers, a copy stream, events both ways
        # (the copy of a buffer waits for the kernels that read it, the kernels for their copy).  Every step = one copy + one
        # batch; the GPU work hides under the link.
        bufs = [shard.buf, torch.zeros_like(shard.buf)]
        copy_s = torch.cuda.Stream(device=dev)
        ev_copied = [torch.cuda.Event(), torch.cuda.Event()]
        ev_done = [torch.cuda.Event(), torch.cuda.Event()]
        for e in ev_done:
            e.record(work)
        state = {"i": 0}

        def issue_copy(i):
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(ev_done[i])
                bufs[i][:shard.n].copy_(host, non_blocking=True)
                ev_copied[i].record(copy_s)

        def step_overlapped():
            i = state["i"]
            issue_copy(1 - i)
            work.wait_event(ev_copied[i])
            shard.buf = bufs[i]
            shard.step(st)
            ev_done[i].record(work)
            state["i"] = 1 - i

        issue_copy(0)
        step_overlapped()
        torch.cuda.synchronize()
        el = wall_steps(torch, step_overlapped, k2)
        ms3 = el / k2 * 1e3
        torch.cuda.synchronize()
        c3 = shard.check()
        assert c3 == c, (c3, c)  # (the batches that came over the link gave what the resident one gave)
        shard.buf = bufs[0]
        out["incl_h2d"] = {"value": round(n_docs / (ms3 / 1e3), 1), "unit": "docs/s", "ms_per_batch": round(ms3, 3),
                           "pcie_floor_ms": round(floor_ms, 3), "pcie_floor_docs_per_s": round(n_docs / (floor_ms / 1e3), 1),
                           "h2d_GBps": round(shard.n / floor_ms / 1e6, 2),
                           "serial_ms_per_batch": round(ms2, 3), "serial_docs_per_s": round(n_docs / (ms2 / 1e3), 1),
                           "note": "pinned host buffer -> HBM copy of the batch inside every step (outputs stay on the device); the copy of "
                                   "batch k + 1 on a second stream under the kernels of batch k (two device input buffers); serial_*: copy "
                                   "then kernels on one stream; pcie_floor: the copy alone"}
        del bufs
    ctx.close()
    return out


def batch_rejected_path(torch, oracle, shard, offs, st, accepted_ms, steps):
    """What a batch costs that the optimistic pipeline does NOT take (the reference isolates a bad document for free: one
    JsonParsingException per parse, SimdJsonParser.java:35-40).  Same documents as the accepted figure, every variant compared per
    document with the oracle (check_batch_all) before it is timed:
      exact           sjmi_parse_batch_device on the unchanged batch (everything queued, the device decides);
      one_bad_doc     ONE document of the batch fails stage 1 (a 0xFF byte inside its last string): optimistic call ->
                      SJMI_ST_REJECTED -> BatchShard.check() makes the call for rejected batches (sjmi_parse_batch_device_rejected:
                      verdicts + the pipeline over a sanitized copy) -- the whole cliff, both calls and the host round trip
                      between them, per batch;
      one_bad_doc_stage2   ONE document fails stage 2 only (its last '}' replaced by ']'): the plain pass is accepted, the token
                      walker declines the document, the exact walker (list mode) reports its error -- no rejection;
      no_separator    the '\\n' behind every document replaced by a space: no control-character separators -> rejected -> the call
                      for rejected batches; no_separator_latched: the steps after that one (BatchShard remembers a rejection for
                      the batch's FORMAT and goes straight to that call)."""
    import numpy as np
    n_docs = shard.n_docs
    res = {}

    def timed(fn):
        fn()
        el = wall_steps(torch, fn, steps)
        return round(el / steps * 1e3, 3)

    def cliff():
        shard.rejected_steps = 0
        shard.format_rejected = False
        shard.step(st)
        torch.cuda.synchronize()
        return shard.check()

    def verified(label, want_failed):
        c = shard.check()
        assert c["failed_documents"] == want_failed and c["host_documents"] == 0, (label, c)
        return check_batch_all(torch, oracle, shard, offs)

    # the accepted step the way the cliffs below are timed: every step synchronised and checked on the host (no two steps in flight)
    res["accepted_checked_each_step_ms"] = timed(cliff)
    # exact call, unchanged batch
    shard.step(st, exact=True)
    torch.cuda.synchronize()
    verified("exact", 0)
    res["exact_ms"] = timed(lambda: shard.step(st, exact=True))
    k = n_docs // 2
    end = int(offs[k + 1])
    # one document fails stage 2 only: "...}\n" -> "...]\n"
    keep = shard.buf[end - 2].clone()
    shard.buf[end - 2] = 0x5D
    shard.rejected_steps = 0
    c = cliff()
    assert getattr(shard, "rejected_steps", 0) == 0, "a stage-2 failure must not reject the batch"
    verified("one_bad_doc_stage2", 1)
    res["one_bad_doc_stage2_ms"] = timed(lambda: shard.step(st))
    shard.buf[end - 2] = keep
    # one document fails stage 1: 0xFF inside the filler string ("...x\"}\n": the byte in front of the closing quote)
    keep = shard.buf[end - 4].clone()
    shard.buf[end - 4] = 0xFF
    cliff()
    assert shard.rejected_steps == 1
    verified("one_bad_doc", 1)
    res["one_bad_doc_ms"] = timed(cliff)
    shard.buf[end - 4] = keep
    # no separators
    sep = torch.from_numpy(np.asarray(offs[1:], dtype=np.int64) - 1).to(shard.buf.device)
    shard.buf[sep] = 0x20
    cliff()
    assert shard.rejected_steps == 1
    verified("no_separator", 0)
    res["no_separator_ms"] = timed(cliff)
    # ... and once BatchShard has latched "rejected for its format": every later step is the call for rejected batches alone
    cliff()
    assert shard.format_rejected
    res["no_separator_latched_ms"] = timed(lambda: shard.step(st))
    verified("no_separator (latched)", 0)
    shard.buf[sep] = 0x0A
    shard.rejected_steps = 0
    shard.format_rejected = False
    shard.step(st)
    torch.cuda.synchronize()
    verified("restored", 0)
    res["accepted_ms"] = accepted_ms
    res["one_bad_doc_over_accepted"] = round(res["one_bad_doc_ms"] / accepted_ms, 2)
    res["one_bad_doc_over_accepted_checked_each_step"] = round(res["one_bad_doc_ms"] / res["accepted_checked_each_step_ms"], 2)
    res["oracle_checked"] = "every variant: all %d documents per document against the oracle before timing" % n_docs
    return res


# ---------------------------------------------------------------------------------------------------------------
# N > 1: the sharded batch
# ---------------------------------------------------------------------------------------------------------------
def memory_system_rate():
    """tools/ubench/load_pattern q -> its three figures (TB/s), or None when the binary is not there / does not run"""
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench", "load_pattern")
    if not os.path.exists(exe):
        return None
    try:
        import torch
        torch.cuda.synchronize()
        torch.cuda.empty_cache()  # (the microbenchmark allocates 5.5 GB of its own)
        out = subprocess.run([exe, "q"], capture_output=True, text=True, timeout=120)
        last = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(last)
        d["what"] = ("tools/ubench/load_pattern.hip on this box, in this run: 1,024 persistent workgroups, a wave reads 4 KiB steps of a 4 GiB buffer "
                     "(k_stage1's load pattern) and stores 0.34 bytes per byte read contiguously in 5.6 KB bursts (twitter.json's indexes: 0.35)")
        return d
    except Exception:  # a yardstick only: never in the way of the line
        return None


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def bench_sharded(args, world):
    """N ranks, one per GPU, over RCCL (torch.distributed backend "nccl").  The path shards by DOCUMENT and the only
    collective is the all_gather of 4 x int64 per rank (north_star).  ONE line, three measurements:
      value            the headline metric of the N = 1 line, weak-scaled: every rank scans ITS shard of a batch of
                       N x `reps` twitter.json documents (reps per rank = 4 GiB) -- stage 1 + the count gather per step;
                       aggregate GB/s of JSON, so that value(N) / (N * value(1)) is the scaling efficiency of the metric;
      batched.weak     configs[3], `--docs` UNIQUE ~1 KB documents PER RANK, sjmi_parse_batch_device + the count gather;
      batched.strong   configs[3], `--docs` documents IN ALL, contiguous byte-balanced ranges (sharding.partition_documents)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import simdjson_java_amd as S
    from simdjson_java_amd import sharding
    import workloads as W
    from oracle import oracle  # checker only

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (test hooks: SJMI_BENCH_BACKEND=gloo + SJMI_BENCH_OVERSUBSCRIBE=1 run the ranks on whatever GPUs exist, e.g. two ranks on
    #  the one GPU of a test box; without them a rank without a GPU of its own is an error, not a silent fallback)
    backend = os.environ.get("SJMI_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < world and os.environ.get("SJMI_BENCH_OVERSUBSCRIBE", "0") != "1":
        print("bench.py: %d ranks but only %d GPU(s) visible -- refusing to run (one rank per GPU)" % (world, ndev), file=sys.stderr)
        sys.exit(3)
    if ndev < world:
        # (the test mode: ranks share a GPU, so no rank's persistent FAST-mode grid is resident as a whole -- every context of this
        #  process starts in ticket mode instead of finding a tripped liveness bound in a result record; one rank per GPU never gets here)
        os.environ["SJMI_TICKET_MODE"] = "1"
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    if dist.get_world_size() != args.gpus:
        print("bench.py: the process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus), file=sys.stderr)
        sys.exit(3)
    work = torch.cuda.Stream(device=dev)
    work.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(work)
    st = work.cuda_stream
    # (anything that might still have to be compiled -- the checker, the document generator; all of it normally travels prebuilt --
    #  is built by rank 0 alone, the others wait: N ranks compiling the same .so at once would corrupt it)
    if rank == 0:
        oracle.build()
        W.build_docgen()
    dist.barrier()
    oracle.build()

    def gather_rows(row):  # the ONLY collective: [world, 4] int64, device tensors (gloo: through the host)
        if backend == "nccl":
            out = torch.empty(world * 4, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(out, row)
            return out.view(world, 4)
        outs = [torch.empty(4, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(outs, row.cpu())
        return torch.stack(outs).to(dev)

    # ---- (A) the headline metric, weak: stage 1 over this rank's `reps` twitter.json documents + the count gather ----
    doc = W.load_twitter()
    idx0, st0 = oracle.stage1(doc)
    assert st0 == 0 and idx0.size == 55263
    reps = args.reps
    buf, n = W.repeat_on_device(doc, reps, dev)
    s_total = idx0.size * reps
    r1 = Stage1Runner(torch, S, dev, work, buf, n, s_total)
    r1.launch()
    torch.cuda.synchronize()
    assert r1.status() == (s_total, 0)
    ok, bad = W.closed_form_ok(r1.out, idx0, len(doc), reps)
    assert ok, "rank %d: GPU indexes differ from the oracle's closed form (copies %d..)" % (rank, bad)
    reps_t = torch.tensor([reps], dtype=torch.int64, device=dev)
    zero_t = torch.zeros(1, dtype=torch.int64, device=dev)
    gathered = None

    def step_a():
        nonlocal gathered
        r1.launch()
        gathered = gather_rows(torch.cat([reps_t, r1.res[0:1], zero_t, r1.res[1:2] & 0xFF]))

    for _ in range(args.warmup + args.preheat):
        step_a()
    r1.ctx.set_profiling(True)
    elapsed_a = wall_steps(torch, step_a, args.steps, dist)
    kern_ms, launches = r1.ctx.kernel_time()
    r1.ctx.set_profiling(False)
    ga = gathered.cpu().numpy()
    assert (ga[:, 1] == s_total).all() and (ga[:, 3] == 0).all() and int(ga[:, 0].sum()) == reps * world
    kernel_ms = kern_ms / max(launches, 1)
    r1.ctx.close()
    del r1, buf
    torch.cuda.empty_cache()

    # ---- (B) configs[3]: the batched path, weak and strong ----
    def run_batch(lo, hi, sample):
        ctx = S.Context(device=dev.index, capacity=1 << 20)
        shard, offs = make_batch_shard(torch, S, W, dev, ctx, lo, hi)
        g = None

        def step():
            nonlocal g
            if backend == "nccl":  # the shard's kernels, then the count gather, all in stream order (sharding.sharded_step)
                g = sharding.sharded_step(shard, st, always_gather=True)
            else:
                shard.step(st)
                g = gather_rows(shard.counts_tensor())

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        c = shard.check()
        assert c["failed_documents"] == 0 and c["host_documents"] == 0 and c["stage1_status"] == 0, c
        checked = check_batch_sample(torch, oracle, shard, offs, sample, seed=20250825 + rank)
        el = wall_steps(torch, step, args.batch_steps, dist)
        alone = None
        if rank == 0:  # the same shard on rank 0 with the other GPUs idle (no collective partner needed: kernels only)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.batch_steps):
                shard.step(st)
            torch.cuda.synchronize()
            alone = (time.perf_counter() - t0) / args.batch_steps * 1e3
        dist.barrier()
        gg = g.cpu().numpy()
        ctx.close()
        return el / args.batch_steps * 1e3, gg, c, checked, alone

    d = args.docs
    w_ms, w_g, w_c, w_chk, w_alone = run_batch(rank * d, (rank + 1) * d, args.sample // max(world, 1))
    lens_all = W.unique_doc_lengths(0, d)
    offs_all = np.concatenate([[0], np.cumsum(lens_all)]).astype(np.uint64)
    lo, hi = sharding.partition_documents(offs_all, world)[rank]
    s_ms, s_g, s_c, s_chk, s_alone = run_batch(lo, hi, args.sample // max(world, 1))
    assert int(s_g[:, 0].sum()) == d and int(w_g[:, 0].sum()) == d * world
    if rank == 0:
        ms = elapsed_a / args.steps * 1e3
        value = world * n * args.steps / elapsed_a / 1e9
        single_docs = d / (w_alone / 1e3)  # rank 0 alone on --docs documents = the one-GPU figure of this run
        batched = {
            "weak": {"value": round(d * world / (w_ms / 1e3), 1), "unit": "docs/s", "ms_per_step": round(w_ms, 4), "scaling": "weak",
                     "documents": d * world, "documents_per_rank": [int(x) for x in w_g[:, 0]],
                     "structurals_per_rank": [int(x) for x in w_g[:, 1]], "rccl_world_size": world,
                     "efficiency_vs_rank0_alone": round(w_alone / w_ms, 4)},
            "strong": {"value": round(d / (s_ms / 1e3), 1), "unit": "docs/s", "ms_per_step": round(s_ms, 4), "scaling": "strong",
                       "documents": d, "documents_per_rank": [int(x) for x in s_g[:, 0]],
                       "structurals_per_rank": [int(x) for x in s_g[:, 1]], "rccl_world_size": world,
                       "rank0_alone_on_its_shard_ms": round(s_alone, 4),
                       "speedup_vs_single_gpu_same_run": round(single_docs and (d / (s_ms / 1e3)) / single_docs, 3)},
            "single_gpu_same_run": {"value": round(single_docs, 1), "unit": "docs/s", "ms_per_batch": round(w_alone, 4),
                                    "documents": d, "note": "rank 0 on its weak shard with the other GPUs idle, kernels only"},
            "oracle_checked_documents_per_rank": w_chk + s_chk,
            "workload": "configs[3]: UNIQUE ~1 KB documents (tools/docgen.c seed 20250825, lengths uniform in [768, 1280] B), per rank "
                        "sjmi_parse_batch_device on its contiguous range (stage 1 with per-document verdicts -> string records -> GPU "
                        "walk), then ONE all_gather of 4 x int64 per rank per step",
        }
        line = {
            "metric": "GB/s JSON scanned (stage-1)", "value": round(value, 2), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic: twitter.json (631,515 B reference fixture) x%d documents per rank (%d B per rank), resident in HBM, "
                    "sharded by document over %d GPUs" % (reps, n, world),
            "config": {"workload": "batched mode, sharded by document: every rank runs stage 1 (UTF-8 validation + structural "
                                   "indexing + uint32 compaction) over its own %d twitter.json documents, then ONE all_gather of "
                                   "{docs, structurals, string bytes, status} per rank per step (the count gather, the only collective); "
                                   "every rank's index array checked against the oracle's closed form before timing" % reps,
                       "bytes_per_gpu": n, "structurals_per_gpu": s_total, "preheat_launches": args.preheat,
                       "sharding": "by document, contiguous", "rccl_world_size": world, "backend": backend,
                       "documents_per_rank": [int(x) for x in ga[:, 0]],
                       "collective": "all_gather_into_tensor of 4 x int64 per rank",
                       "batched_weak_docs_per_s": batched["weak"]["value"], "batched_strong_docs_per_s": batched["strong"]["value"],
                       "batched_single_gpu_docs_per_s": batched["single_gpu_same_run"]["value"],
                       "batched_documents": d},
            "backend": backend + (" (RCCL)" if backend == "nccl" else ""),
            "roofline": roofline(n + 4 * (s_total + 1), kernel_ms, n, "k_stage1 (rank 0)", launches,
                                 timed_region="the K timed steps on rank 0 (HIP events on its launch stream)",
                                 all_ranks_achieved=round(value, 2), all_ranks_peak=HBM_PEAK_GBS * world,
                                 all_ranks_frac=round(value / (HBM_PEAK_GBS * world), 4)),
            "batched": batched,
        }
        print(json.dumps(line))
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preheat", type=int, default=40,
                    help="N=1: further untimed launches before the timed steps (the clock governor needs ~25 ms of back-to-back "
                         "launches to settle; the cold figure is reported beside the settled one in `roofline`)")
    ap.add_argument("--reps", type=int, default=6801, help="N=1: copies of twitter.json (6801 = 4 GiB north-star, 1024 = configs[1])")
    ap.add_argument("--tile-steps", type=int, default=0, help="force the chain granule = N x 4 KiB: 1, 2 or 4 (0 = auto)")
    ap.add_argument("--docs", type=int, default=1000000, help="documents of the configs[3] batch")
    ap.add_argument("--pool", type=int, default=4000, help="unique documents of the bounded CPU-baseline sample of configs[3]")
    ap.add_argument("--batch-steps", type=int, default=20, help="N=1: timed steps of the configs[3] extra")
    ap.add_argument("--batch-preheat", type=int, default=20, help="N=1: untimed steps of the configs[3] extra in front of the timed ones (disclosed)")
    ap.add_argument("--batch-accepted-only", action="store_true",
                    help="N=1: the configs[3] extra without its rejected-path and H2D variants (the counter passes of tools/prof_round.sh: "
                         "the last dispatches of every kernel are then those of an accepted step)")
    ap.add_argument("--no-extras", action="store_true", help="N=1: only the primary workload")
    ap.add_argument("--sections", default="x1024,unescape,synth,batch,parse,select,trees",
                    help="N=1: which extras to run (comma list of x1024, unescape, synth, batch, parse, select); the profiling passes run one each")
    ap.add_argument("--skip-main-timing", action="store_true",
                    help="N=1, profiling passes of an extra only: check the primary workload once, do not time it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true",
                    help="N=1: take the sharded branch anyway -- init_process_group('nccl'), the batch on one rank, the count gather as a "
                         "real RCCL all_gather_into_tensor in a group of one (prints the SCALE-style line instead of the BENCH line)")
    ap.add_argument("--sample", type=int, default=10000, help="documents of the configs[3] batch checked against the oracle before timing")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # launched plainly (`python bench.py --gpus N`): become the N ranks -- one process per GPU under torch.distributed.run
            # -- instead of quietly measuring one GPU; fewer than N GPUs is an error
            import torch
            if torch.cuda.device_count() < args.gpus and os.environ.get("SJMI_BENCH_OVERSUBSCRIBE", "0") != "1":
                print("bench.py: --gpus %d but only %d GPU(s) visible -- refusing to run" % (args.gpus, torch.cuda.device_count()),
                      file=sys.stderr)
                sys.exit(3)
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.call(cmd, env=env))
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            print("bench.py: WORLD_SIZE (%d) != --gpus (%d) -- refusing to run" % (world, args.gpus), file=sys.stderr)
            sys.exit(3)
    if world > 1:
        bench_sharded(args, world)
    elif args.sharded:
        # a process group of ONE rank over RCCL: what torch.distributed.run would have put into the environment
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("LOCAL_RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        bench_sharded(args, 1)
    else:
        bench_single(args)


if __name__ == "__main__":
    main()
