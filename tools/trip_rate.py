#!/usr/bin/env python
"""FAST-mode liveness under contention: how often does a FAST k_stage1 launch trip its spin bound
(SJMI_ST_INTERNAL) when other persistent kernels compete for the GPU, and what does the SAFE re-run cost?

  victim     this process: N FAST launches of stage 1 over twitter x64 (40 MB), every result record read back (no auto-SAFE:
             a tripped bound is COUNTED, then the launch is repeated in SAFE mode and timed);
  rccl       (optional) a thread of this process: all_gather_into_tensor in a loop over a real "nccl" group of one rank
             on its own stream -- the kernel RCCL launches on the device;
  neighbour  (optional) a second PROCESS looping k_strings (another persistent kernel with a scanner chain) over its own
             40 MB document.

usage: trip_rate.py <launches> [rccl] [neighbour | neighbours=N]      -> one JSON line   (round 5: N contending processes)
       trip_rate.py neighbour-worker <seconds>        (internal)"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import simdjson_java_amd as S
import workloads as W
from oracle import oracle as O

REPS = 64


def neighbour(seconds):
    doc = W.load_twitter()
    idx0, _ = O.stage1(doc)
    dev = torch.device("cuda", 0)
    buf, n = W.repeat_on_device(doc, REPS, dev)
    cap = idx0.size * REPS + 1
    out = torch.empty(cap, dtype=torch.int32, device=dev)
    res = torch.zeros(2, dtype=torch.int64, device=dev)
    ctx = S.Context(0, 1 << 20)
    ctx.set_auto_safe(True)
    st = torch.cuda.Stream()
    ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), st.cuda_stream)
    sb_cap = n + 4 * cap + 64
    sb = torch.zeros(sb_cap, dtype=torch.uint8, device=dev)
    ures = torch.zeros(3, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    print("neighbour ready", flush=True)
    t0, k = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(8):
            ctx.unescape_device(buf.data_ptr(), n, out.data_ptr(), idx0.size * REPS, sb.data_ptr(), sb_cap, ures.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        k += 8
    print("neighbour done: %d string passes" % k, flush=True)
    ctx.close()


def main():
    if sys.argv[1] == "neighbour-worker":
        return neighbour(float(sys.argv[2]))
    launches = int(sys.argv[1])
    with_rccl, with_nb = "rccl" in sys.argv[2:], "neighbour" in sys.argv[2:]
    n_nb = 1 if with_nb else 0
    for a in sys.argv[2:]:
        if a.startswith("neighbours="):
            n_nb = int(a.split("=")[1])
            with_nb = n_nb > 0
    doc = W.load_twitter()
    idx0, _ = O.stage1(doc)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    buf, n = W.repeat_on_device(doc, REPS, dev)
    cap = idx0.size * REPS + 1
    out = torch.empty(cap, dtype=torch.int32, device=dev)
    res = torch.zeros(2, dtype=torch.int64, device=dev)
    ctx = S.Context(0, 1 << 20)
    work = torch.cuda.Stream()
    stop = threading.Event()
    gathers = [0]
    th = None
    if with_rccl:
        import socket
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

        def loop():
            s2 = torch.cuda.Stream()
            row = torch.arange(4, dtype=torch.int64, device=dev)
            o = torch.empty(4, dtype=torch.int64, device=dev)
            with torch.cuda.stream(s2):
                while not stop.is_set():
                    for _ in range(32):
                        dist.all_gather_into_tensor(o, row)
                    s2.synchronize()
                    gathers[0] += 32
        th = threading.Thread(target=loop, daemon=True)
        th.start()
    nbs = []
    for _ in range(n_nb):
        nb = subprocess.Popen([sys.executable, os.path.abspath(__file__), "neighbour-worker", str(max(20.0, launches * 0.0006 * max(1, n_nb)))],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        nbs.append(nb)
    for nb in nbs:  # "neighbour ready" (after its imports and first launch)
        while True:
            l = nb.stdout.readline()
            if "ready" in l or not l:
                break
    trips, wrong, safe_ms = 0, 0, []
    t0 = time.time()
    for it in range(launches):
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), work.cuda_stream)
        work.synchronize()
        r = res.cpu().numpy()
        st = int(r[1]) & 0xFFFFFFFF
        if st & 0x200:  # SJMI_ST_INTERNAL: the spin bound tripped -> what the caller does: the same launch in SAFE mode
            trips += 1
            ctx.debug_set_flags(0x100)
            t1 = time.perf_counter()
            ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), work.cuda_stream)
            work.synchronize()
            safe_ms.append((time.perf_counter() - t1) * 1e3)
            ctx.debug_set_flags(0)
            r = res.cpu().numpy()
            st = int(r[1]) & 0xFFFFFFFF
        if int(r[0]) != idx0.size * REPS or st != 0:
            wrong += 1
    el = time.time() - t0
    stop.set()
    if th:
        th.join(timeout=10)
    ok, _ = W.closed_form_ok(out, idx0, len(doc), REPS)
    nb_out = []
    for nb in nbs:
        try:
            o, _ = nb.communicate(timeout=300)
            nb_out.append(o.strip().splitlines()[-1] if o.strip() else "")
        except subprocess.TimeoutExpired:
            nb.kill()
            nb_out.append("TIMEOUT")
    print(json.dumps({"launches": launches, "document_MB": n // 1000000, "rccl_all_gather_loop": with_rccl, "rccl_gathers": gathers[0],
                      "contending_string_pass_processes": n_nb, "neighbours": nb_out or None,
                      "tripped_spin_bounds": trips, "wrong_results": wrong, "final_indexes_ok": bool(ok),
                      "safe_rerun_ms": [round(x, 2) for x in safe_ms[:8]], "ms_per_launch_incl_sync": round(el / launches * 1e3, 4)}))
    ctx.close()


if __name__ == "__main__":
    main()
