"""CPU fuzz of the device block algebra (simdjson-java_amd/csrc/sj_block.h) against the oracle.

The header is shared verbatim between the HIP kernels and this host build, so the plane
classification, halo carries, UTF-8 plane algebra and the parity flip are checked here without a
GPU; the GPU tests then cover the transposition intrinsics, the scans and the tile look-back."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT, load_fixture

SIM_DIR = os.path.join(ROOT, "tests", "host_sim")


@pytest.fixture(scope="module")
def sim():
    so = os.path.join(SIM_DIR, "libsim.so")
    src = os.path.join(SIM_DIR, "sim.cpp")
    hdr = os.path.join(ROOT, "simdjson-java_amd", "csrc", "sj_block.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    lib = C.CDLL(so)
    lib.sim_stage1.restype = C.c_int
    lib.sim_stage1.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]

    lib.sim_masks.restype = C.c_int
    lib.sim_masks.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]

    def masks(data):
        a = np.frombuffer(bytes(data) + b"\0" * 64, dtype=np.uint8)
        m = np.zeros((len(data) // 64 + 1, 6), dtype=np.uint64)
        assert lib.sim_masks(a.ctypes.data, len(data), m.ctypes.data) == 0
        return m

    def run(data):
        a = np.frombuffer(bytes(data) + b"\0" * 64, dtype=np.uint8)
        n = len(data)
        idx = np.empty(n + 2, dtype=np.uint32)
        cnt = C.c_uint64(0)
        st = C.c_uint32(0)
        assert lib.sim_stage1(a.ctypes.data, n, idx.ctypes.data, n + 2, C.addressof(cnt), C.addressof(st)) == 0
        return idx[:cnt.value].copy(), st.value
    run.masks = masks
    return run


def _check(sim, d):
    i1, s1 = O.stage1(d)
    i2, s2 = sim(d)
    assert s1 == s2, (s1, s2, d[:200].hex())
    assert np.array_equal(i1, i2), d[:200].hex()


def test_files(sim):
    for name in ("twitter.json", "github_events.json", "wide_bench.json", "malformed.txt"):
        _check(sim, load_fixture(name))


def test_fuzz_json_like(sim):
    rng = random.Random(5)
    alphabet = b'\\\\\\"""{}[]:, \t\n\r\x0c\x1a\x01abc019.-e'
    for it in range(4000):
        n = rng.choice([0, 1, 63, 64, 65, 127, 128, 129, rng.randint(0, 700)])
        mode = it % 4
        if mode == 0:
            d = bytes(rng.choice(alphabet) for _ in range(n))
        elif mode == 1:
            d = bytes(rng.choice(b'\\"a ') for _ in range(n))
        elif mode == 2:
            d = b"a" * rng.randint(0, 70) + b"\\" * rng.randint(1, 300) + rng.choice([b'"', b"x", b""]) + b'"x' * rng.randint(0, 40)
        else:
            d = bytes(rng.getrandbits(8) for _ in range(n))
        _check(sim, d)


def test_fuzz_utf8(sim):
    rng = random.Random(6)
    interesting = [0x00, 0x22, 0x5C, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC1, 0xC2, 0xDF, 0xE0, 0xE1,
                   0xEC, 0xED, 0xEE, 0xEF, 0xF0, 0xF1, 0xF3, 0xF4, 0xF5, 0xF7, 0xF8, 0xFF, 0x41]
    valid_chars = ["a", "é", "€", "한", "ࠀ", "퟿", "", "￿", "😀", "\U00010000", "\U0010ffff", "ࠀ", "ก"]
    for it in range(6000):
        if it % 2:
            d = bytes(rng.choice(interesting) for _ in range(rng.randint(0, 200)))
        else:
            s = "".join(rng.choice(valid_chars) for _ in range(rng.randint(0, 90))).encode()
            pad = b"x" * rng.randint(0, 66)
            base = bytearray(pad + s)
            if it % 4 == 0 and base:
                for _ in range(rng.randint(1, 2)):
                    base[rng.randrange(len(base))] = rng.choice(interesting)
            if it % 8 == 2 and base:  # truncate mid-sequence
                base = base[:rng.randrange(len(base)) + 1]
            d = bytes(base)
        _check(sim, d)


MASK_NAMES = ["escaped", "quote", "inString", "op", "whitespace", "structurals"]


def _check_masks(sim, d):
    _, _, want = O.index_blocks(d, want_masks=True)
    got = sim.masks(d)
    assert got.shape == want.shape
    if not np.array_equal(got, want):
        b, k = [int(x[0]) for x in np.nonzero(got != want)]
        raise AssertionError("block %d mask %s: got %016x want %016x" % (b, MASK_NAMES[k], int(got[b, k]), int(want[b, k])))


def test_reference_bitmasks(sim):
    """The six per-block masks of StructuralIndexer.java:210-252 rebuilt from the kernel's pot / sm0 formulation
    (sj_reference_masks, the function csrc/masks.hip runs on the device) equal the oracle's line-by-line restatement:
    reference files, the reference's StructuralIndexerTest inputs, backslash runs across block boundaries, fuzz."""
    from tests.golden import vectors as V
    for name in ("twitter.json", "github_events.json", "wide_bench.json", "malformed.txt"):
        _check_masks(sim, load_fixture(name))
    for case in V.STRUCTURAL_INDEXER:
        _check_masks(sim, case[1])
    rng = random.Random(15)
    alphabet = b'\\\\\\"""{}[]:, \t\n\r\x0c\x1a\x01abc019.-e\xc3\xa9'
    for it in range(3000):
        n = rng.choice([0, 1, 63, 64, 65, 127, 128, 129, rng.randint(0, 900)])
        mode = it % 4
        if mode == 0:
            d = bytes(rng.choice(alphabet) for _ in range(n))
        elif mode == 1:
            d = bytes(rng.choice(b'\\"a ') for _ in range(n))
        elif mode == 2:
            d = b"a" * rng.randint(0, 70) + b"\\" * rng.randint(1, 300) + rng.choice([b'"', b"x", b""]) + b'"x' * rng.randint(0, 40)
        else:
            d = bytes(rng.getrandbits(8) for _ in range(n))
        _check_masks(sim, d)


def test_block32_equals_block():
    """sj_block32.h (the block algebra in fast-class VALU instructions: 32-bit halves, explicit three-input boolean functions,
    butterfly-only transposition) against sj_block.h and the bit-loop transposition: 600 k biased random blocks with every
    combination of carries, tails of every length, with and without the tape-word counts and the ASCII shortcut."""
    import ctypes as C
    so = os.path.join(SIM_DIR, "libblock32.so")
    src = os.path.join(SIM_DIR, "block32.cpp")
    deps = [src] + [os.path.join(ROOT, "simdjson-java_amd", "csrc", h) for h in ("sj_block.h", "sj_block32.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    L = C.CDLL(so)
    at = C.c_uint64(0)
    for seed in (1, 2, 3):
        rc = L.block32_fuzz(C.c_uint64(seed), C.c_uint64(200000), C.byref(at))
        assert rc == 0, "check %d failed in round %d (seed %d)" % (rc, at.value, seed)
