"""Workloads of BASELINE.json `configs`, built on the device for bench.py and the full-scale GPU tests.

  configs[1]  twitter.json x reps byte-concatenated (x1024 = 646,671,360 B; x6801 = 4,294,933,515 B, the north-star
              "4 GiB concatenated twitter.json": the largest multiple that keeps uint32 indexes)
  configs[2]  4 GiB synthetic JSON (tools/synth.synth_tile_spec: a 16 MiB tile x255; 50 % of the bytes in string literals, 10 % of
              the string characters escapes, 10 % non-ASCII -- SURVEY.md 8(d); the measured shares are printed with the number)
  configs[3]  1,000,000 ~1 KB documents packed NDJSON-style (a pool of unique documents from tools/synth.small_docs,
              repeated in order), with their u64 offsets table
Every builder returns what a closed-form parity check needs (the oracle's result for ONE tile / ONE pool)."""
import gzip
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "tools") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tools"))

TWITTER_4G_REPS = 6801  # 6801 * 631,515 = 4,294,933,515 < 2^32 <= 6802 * 631,515


def load_twitter():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "data", "twitter.json.gz"), "rb") as f:
        return f.read()


def repeat_on_device(tile, reps, dev, pad=128):
    """tile x reps in HBM, 16-byte aligned, `pad` zero bytes behind it -> (uint8 tensor, n)."""
    import torch
    n = len(tile) * reps
    buf = torch.zeros(n + pad, dtype=torch.uint8, device=dev)
    buf[:n] = torch.frombuffer(bytearray(tile), dtype=torch.uint8).to(dev).repeat(reps)
    return buf, n


def closed_form_ok(out, idx0, n0, reps, chunk=64):
    """index[k * S + j] == k * n0 + idx0[j] for the whole uint32 array `out` (a device int32 tensor), compared on the
    device in chunks of `chunk` copies; -> (ok, first bad copy or -1)."""
    import torch
    dev = out.device
    s = idx0.size
    base = torch.from_numpy(idx0.astype(np.int64)).to(dev)
    for k0 in range(0, reps, chunk):
        k1 = min(reps, k0 + chunk)
        want = (base[None, :] + (torch.arange(k0, k1, device=dev, dtype=torch.int64) * n0)[:, None]).flatten()
        got = out[k0 * s:k1 * s].to(torch.int64) & 0xFFFFFFFF
        if not torch.equal(got, want):
            return False, k0
    return True, -1


def synth_tile(target_bytes=16 << 20):
    """the configs[2] tile as SURVEY.md 8(d) specifies it (16 MiB, 50 % of the bytes inside string literals)"""
    import synth
    return synth.synth_tile_spec(target_bytes=target_bytes)


def measured_fractions(tile, masks):
    """what the tile really is (from the oracle's masks): share of bytes inside string literals, of escape sequences among
    the string characters, of non-ASCII bytes"""
    a = np.frombuffer(tile, dtype=np.uint8)
    in_str = int(sum(bin(int(x)).count("1") for x in masks[:, 2]))
    esc = int(sum(bin(int(x)).count("1") for x in masks[:, 0]))  # escaped characters = escape sequences
    return {"bytes_in_string_literals": round(in_str / len(tile), 4), "escape_sequences_per_string_byte": round(esc / max(in_str, 1), 4),
            "non_ascii_bytes": round(float((a >= 0x80).mean()), 4)}


def small_doc_pool(unique=4000, same_schema=False):
    """-> (list of documents, unit = the pool packed with one '\\n' behind each document, lens incl. the separator)"""
    import synth
    docs = synth.small_docs(n=unique, same_schema=same_schema)
    unit = b"".join(d + b"\n" for d in docs)
    lens = np.array([len(d) + 1 for d in docs], dtype=np.uint64)
    return docs, unit, lens


def batch_offsets(lens, reps):
    return np.concatenate([[0], np.cumsum(np.tile(lens, reps))]).astype(np.uint64)


# ---- configs[3] as SURVEY.md 8(d) states it: 1,000,000 UNIQUE documents (tools/docgen.c) ----------------------------------
_DOCGEN = None
DOCGEN_SEED = 20250825


def build_docgen(force=False):
    """gcc -> tools/libdocgen.so (in-tree, travels to the GPU box; __graft_entry__.build() calls this)"""
    import subprocess
    src, lib = os.path.join(ROOT, "tools", "docgen.c"), os.path.join(ROOT, "tools", "libdocgen.so")
    stale = lambda: not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src)
    if force or stale():
        import fcntl
        # (compiled into a temporary file and renamed into place under a lock: several ranks may get here at once)
        with open(os.path.join(ROOT, "tools", ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or stale():
                tmp = "%s.tmp.%d" % (lib, os.getpid())
                subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", tmp, src])
                os.replace(tmp, lib)
    return lib


def _docgen():
    global _DOCGEN
    if _DOCGEN is None:
        import ctypes
        L = ctypes.CDLL(os.environ.get("SJMI_DOCGEN_LIB") or build_docgen())  # (SJMI_DOCGEN_LIB: an experiment build, e.g. -DDOC_SCALE=4)
        L.docgen_lengths.argtypes = [ctypes.c_uint64] * 3 + [ctypes.c_void_p]
        L.docgen_lengths.restype = None
        L.docgen_fill.argtypes = [ctypes.c_uint64] * 3 + [ctypes.c_void_p, ctypes.c_void_p]
        L.docgen_fill.restype = None
        _DOCGEN = L
    return _DOCGEN


def unique_doc_lengths(first, n, seed=DOCGEN_SEED, threads=None):
    """lengths (incl. the '\\n' separator) of documents [first, first + n) of the set; a pure function of (seed, k)"""
    from concurrent.futures import ThreadPoolExecutor
    L = _docgen()
    lens = np.zeros(n, dtype=np.uint64)
    threads = threads or min(32, os.cpu_count() or 1)
    step = max(1, (n + threads - 1) // threads)
    with ThreadPoolExecutor(threads) as ex:  # (ctypes releases the GIL)
        list(ex.map(lambda a: L.docgen_lengths(seed, first + a, min(step, n - a), lens[a:].ctypes.data), range(0, n, step)))
    return lens


def unique_docs(first, n, seed=DOCGEN_SEED, threads=None, out=None):
    """documents [first, first + n) packed with one '\\n' behind each -> (uint8 array, u64 offsets[n + 1] from 0).
    `out` (optional): a writable uint8 array to fill (e.g. the numpy view of a pinned torch tensor)."""
    from concurrent.futures import ThreadPoolExecutor
    L = _docgen()
    lens = unique_doc_lengths(first, n, seed, threads)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    buf = out if out is not None else np.empty(int(offs[-1]), dtype=np.uint8)
    assert buf.size >= int(offs[-1])
    threads = threads or min(32, os.cpu_count() or 1)
    step = max(1, (n + threads - 1) // threads)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda a: L.docgen_fill(seed, first + a, min(step, n - a), buf.ctypes.data, offs[a:].ctypes.data), range(0, n, step)))
    return buf[:int(offs[-1])], offs
