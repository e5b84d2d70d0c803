#!/usr/bin/env python
"""Per-tile timeline of k_stage1 (experiments only).  Builds a -DSJMI_TRACE copy of the library, runs
twitter.json x reps, and prints how long tiles spend in each phase and how phases line up per CU."""
import ctypes as C, gzip, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import simdjson_java_amd as S
import simdjson_java_amd.binding as B

def build_trace_lib():
    out = os.path.join(ROOT, "tools", "libsjmi_trace.so")
    srcs = [os.path.join(ROOT, "simdjson-java_amd", "csrc", s) for s in B.SOURCES]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSJMI_TRACE",
                               "-I", os.path.join(ROOT, "include")] + srcs + ["-o", out])
    return out

if __name__ == "__main__":
    path = build_trace_lib()
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        sys.exit(0)
    B._LIB = path
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
    n = len(doc) * reps
    buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
    buf[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda().repeat(reps)
    cap = 55263 * reps + 1
    out = torch.empty(cap, dtype=torch.int32, device="cuda")
    res = torch.zeros(2, dtype=torch.int64, device="cuda")
    ctx = S.Context(0, 1 << 20)
    ctx.set_tile_steps(steps)
    ctx.set_tile_mode(mode)
    work = torch.cuda.Stream(); torch.cuda.synchronize()
    for _ in range(3):
        ctx.stage1_device(buf.data_ptr(), n, out.data_ptr(), cap, res.data_ptr(), work.cuda_stream)
    torch.cuda.synchronize()
    tiles = (n // 64 + 1 + 64 * steps - 1) // (64 * steps)
    tr = np.zeros((tiles, 6), dtype=np.uint64)
    L = B.lib()
    L.sjmi_debug_read_ws.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    rc = L.sjmi_debug_read_ws(ctx._h, tr.ctypes.data, 640 + 16 * tiles, tr.nbytes)
    assert rc == 0, rc
    t = tr[:, :5].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0  # 100 MHz -> us
    hw = tr[:, 5]
    cu = (hw & 0xFFFFFFFF).astype(np.int64)
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    cu_id = (cu >> 8) & 0xF; sh_id = (cu >> 12) & 1; se_id = (cu >> 13) & 7
    print("tiles %d  kernel span %.1f us" % (tiles, t[:, 4].max()))
    names = ["classify+publish", "publish -> look-back start", "look-back", "expand+store"]
    def stats(name, d):
        print("%-28s mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (name, d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
    for k in range(4):
        stats(names[k], t[:, k + 1] - t[:, k])
    stats("tile life", t[:, 4] - t[:, 0])
    cummax = np.maximum.accumulate(t[:, 1])
    need = np.concatenate([[0.0], cummax[:-1]])
    stats("look-back start - all predecessors published (slack, <0 = must wait)", t[:, 2] - need)
    np.save(os.path.join(ROOT, "gpurun_out", "trace.npy"), tr)
    key = xcc * 1000 + se_id * 100 + sh_id * 10 + cu_id
    print("distinct (xcc,se,sh,cu): %d" % len(np.unique(key)))
    for kk in np.unique(key)[:2]:
        sel = np.where(key == kk)[0][:30]
        print("CU key", kk)
        for j in sel:
            print("  tile %6d  start %8.2f  classify %6.2f  gap %6.2f  lookback %6.2f  emit %6.2f  end %8.2f" % (
                j, t[j, 0], t[j, 1] - t[j, 0], t[j, 2] - t[j, 1], t[j, 3] - t[j, 2], t[j, 4] - t[j, 3], t[j, 4]))
