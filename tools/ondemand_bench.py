"""Parse-and-select of twitter.json (the reference's headline benchmark shape) through the public C ABI, user code in C++
(tools/ondemand_bench.cpp): full parse + JsonValue walk, on-demand cursor."""
import ctypes as C
import gzip
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import simdjson_java_amd as S  # noqa: E402


def build_bench_lib():
    """g++ only (no GPU needed): built by __graft_entry__.build() so that it travels with the tree; rebuilt here when stale"""
    src = os.path.join(ROOT, "tools", "ondemand_bench.cpp")
    so = os.path.join(ROOT, "tools", "libondemand_bench.so")
    libdir = os.path.join(ROOT, "simdjson-java_amd")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        # ($ORIGIN-relative rpath: the tree is copied to another place on the GPU box)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src, "-L" + libdir, "-lsjmi",
                               "-Wl,-rpath,$ORIGIN/../simdjson-java_amd"])
    return so


def load_bench_lib():
    so = build_bench_lib()
    S.lib()  # libsjmi.so first (with torch's HIP runtime, see binding.lib)
    L = C.CDLL(so)
    L.odb_run.restype = C.c_int
    L.odb_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.odb_parse_batch.restype = C.c_int
    L.odb_parse_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]
    return L


def measure(doc, iters=300, gpu_walk=None):  # None: the library places stage 2 of the full parse by document size
    """-> {mode: ms per parse-and-select}; asserts the 86 selected users of twitter.json"""
    L = load_bench_lib()
    p = S.SimdJsonParser(capacity=len(doc) + 64, gpu_walk=gpu_walk)
    buf = (C.c_uint8 * len(doc)).from_buffer_copy(doc)
    out = {}
    for mode, name in ((0, "full_parse_then_select"), (1, "on_demand_scan")):
        secs, sel, nbytes = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        best = 1e9
        for rep in range(4):  # first round = warm-up
            rc = L.odb_run(p._h, buf, len(doc), mode, iters, C.byref(secs), C.byref(sel), C.byref(nbytes))
            if rc:
                raise SystemExit("odb_run mode %d failed: %d" % (mode, rc))
            if rep:
                best = min(best, secs.value / iters * 1e3)
        out[name] = {"ms": round(best, 4), "ops_per_s": round(1e3 / best, 1), "selected": sel.value}
    p.close()
    return out


if __name__ == "__main__":
    doc = gzip.open(os.path.join(ROOT, "tests/golden/data/twitter.json.gz")).read()
    for k, v in measure(doc).items():
        print(k, v)
