"""GPU parity tests of the fused stage-1 kernel, through the C ABI (include/sjmi.h), against the
CPU oracle on the same seeded inputs.  Bit-exact: indexes, sentinel, count and the three status bits."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import load_fixture
from tests.golden import vectors as V

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fast", "ticket"])
def ctx(request):
    """Both tile-assignment modes of the kernel: tile = workgroup index (default) and atomic ticket (safe)."""
    import simdjson_java_amd as S
    c = S.Context(device=0, capacity=96 * 1024 * 1024)
    c.set_tile_mode(request.param == "ticket")
    yield c
    c.close()


def _check(ctx, d, length=None):
    want_idx, want_st = O.stage1(d, length)
    got_idx, got_st = ctx.stage1(d, length)
    assert got_st == want_st, (got_st, want_st, bytes(d[:120]).hex())
    assert got_idx.size == want_idx.size, (got_idx.size, want_idx.size)
    if not np.array_equal(got_idx, want_idx):
        bad = int(np.nonzero(got_idx != want_idx)[0][0])
        raise AssertionError("first index mismatch at %d: got %d want %d" % (bad, got_idx[bad], want_idx[bad]))


def test_native_library_is_loaded():
    import simdjson_java_amd as S
    import os
    assert os.path.exists(S.lib_path())
    assert S.lib().sjmi_version().startswith(b"sjmi")


def test_transpose_selftest(ctx):
    assert ctx.selftest() == 0


@pytest.mark.parametrize("case", V.STRUCTURAL_INDEXER, ids=[c[0] for c in V.STRUCTURAL_INDEXER])
def test_reference_structural_indexer_vectors(ctx, case):
    import simdjson_java_amd as S
    name, data, want_idx, want_msg, cite = case
    idx, st = ctx.stage1(data)
    if want_msg is not None:
        assert S.status_message(st) == want_msg, cite
    else:
        assert st == 0 and idx.tolist() == want_idx, cite
    _check(ctx, data)


@pytest.mark.parametrize("steps", [1, 2, 4])
@pytest.mark.parametrize("name", ["twitter.json", "github_events.json", "wide_bench.json", "malformed.txt"])
def test_reference_files(ctx, name, steps):
    ctx.set_tile_steps(steps)
    try:
        d = load_fixture(name)
        _check(ctx, d)
        if name in V.FILES:
            idx, st = ctx.stage1(d)
            assert idx.size == V.FILES[name][1] and st == 0
    finally:
        ctx.set_tile_steps(0)


def test_len_shorter_than_buffer_is_invisible(ctx):
    d = b'[1,2,3]"\\{{{{\xff\xfe' * 40
    for n in (0, 1, 7, 8, 9, 63, 64, 65, 100, 400):
        _check(ctx, d, n)


@pytest.mark.parametrize("steps", [1, 4])
def test_fuzz_small(ctx, steps):
    ctx.set_tile_steps(steps)
    try:
        rng = random.Random(100 + steps)
        alphabet = b'\\\\\\"""{}[]:, \t\n\r\x0c\x1a\x01abc019.-e\xc3\xa9'
        for it in range(300):
            n = rng.choice([0, 1, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, rng.randint(0, 3000)])
            mode = it % 5
            if mode == 0:
                d = bytes(rng.choice(alphabet) for _ in range(n))
            elif mode == 1:
                d = bytes(rng.choice(b'\\"a ') for _ in range(n))
            elif mode == 2:
                d = b"a" * rng.randint(0, 70) + b"\\" * rng.randint(1, 300) + rng.choice([b'"', b"x", b""]) + b'"x' * rng.randint(0, 40)
            elif mode == 3:
                d = bytes(rng.getrandbits(8) for _ in range(n))
            else:
                s = "".join(rng.choice(["a", "é", "€", "한", "😀", '"', "\\\\", " ", ","]) for _ in range(n // 2))
                d = s.encode()
            _check(ctx, d)
    finally:
        ctx.set_tile_steps(0)


def _json_like(rng, n):
    """Dense synthetic JSON-ish text with strings, escapes, non-ASCII and numbers."""
    parts = []
    size = 0
    words = ['"name"', '"va\\"lue"', '"x\\\\"', '"é€한😀"', "12345", "-1.5e10", "true", "false", "null", '"\\u00e9\\ud83d\\ude00"',
             '"' + "abc def " * 9 + '"', '""']
    while size < n:
        k = rng.random()
        if k < 0.5:
            p = rng.choice(words)
        elif k < 0.8:
            p = rng.choice(["{", "}", "[", "]", ":", ",", " ", "\n", "  "])
        else:
            p = '"' + "".join(rng.choice("abcdefghij klmnop") for _ in range(rng.randint(0, 200))) + '"'
        parts.append(p)
        size += len(p)
    return "".join(parts).encode()[:n]


@pytest.mark.parametrize("steps", [1, 2, 4])
def test_fuzz_tile_boundaries(ctx, steps):
    """Sizes around multiples of the tile (steps * 16 KiB) with strings / escapes / multi-byte
    characters straddling block, wave, workgroup and tile boundaries."""
    ctx.set_tile_steps(steps)
    try:
        rng = random.Random(7 + steps)
        tile = steps * 16384
        for it in range(24):
            n = rng.choice([tile - 1, tile, tile + 1, 2 * tile - 64, 2 * tile + 63, 3 * tile + rng.randint(-70, 70),
                            5 * tile + rng.randint(0, 5000)])
            d = bytearray(_json_like(rng, n))
            # plant hazards exactly at tile / wave boundaries
            for b in (tile, 2 * tile, 4096, 8192, tile + 4096):
                if b + 8 < len(d) and b >= 8:
                    kind = rng.randrange(6)
                    if kind == 0:
                        d[b - 1:b + 1] = b'\\"'
                    elif kind == 1:
                        d[b - 3:b + 1] = b'\\\\\\"'
                    elif kind == 2:
                        d[b - 2:b + 2] = "😀".encode()
                    elif kind == 3:
                        d[b - 1:b + 2] = "€".encode()
                    elif kind == 4:
                        d[b - 5:b + 5] = b"\\" * 10
                    else:
                        d[b - 1:b + 1] = b'""'
            _check(ctx, bytes(d))
    finally:
        ctx.set_tile_steps(0)


@pytest.mark.parametrize("steps", [1, 2, 4])
def test_scanner_window_boundaries(ctx, steps):
    """Granule counts around the scanner's windows of 256 granules (and the four-wave round of 1024): the last window
    may be full, one granule long, or shared; hazards planted on the granule boundaries next to them."""
    ctx.set_tile_steps(steps)
    try:
        rng = random.Random(70 + steps)
        g = steps * 4096
        counts = [255, 256, 257, 1024, 1025] if steps == 1 else [256, 257]
        for ngran in counts:
            for delta in (0, 1):
                n = ngran * g - delta
                d = bytearray(_json_like(rng, n))
                for b in (255 * g, 256 * g, 257 * g, 512 * g, 1024 * g, n - g):
                    if 8 <= b and b + 8 < len(d):
                        d[b - 1:b + 1] = rng.choice([b'\\"', b'""', b'",'])
                _check(ctx, bytes(d))
    finally:
        ctx.set_tile_steps(0)


def test_adversarial_runs(ctx):
    n = 200 * 1024
    for d in (b'"' * n, b"\\" * n, b"\\" * (4 * 1024 * 1024 + 3) + b'"x"', b"\\" * (n - 1) + b'"', b'"' + b"\\" * (n - 3) + b'"x', b"[" * n, b" " * n, b"a" * n,
              b'"' + b"\x01" * n + b'"', ("é" * (n // 2)).encode(), ("€" * (n // 3)).encode()[:-1], b"\xf0\x9f\x98" * 1000):
        _check(ctx, d)


def test_many_tiles_parity_chain(ctx):
    """A single string opened in tile 0 and closed ~40 MB later: every tile in between must see
    parity 1 through the look-back chain; then a second document half full of lone quotes."""
    body = (b'{"k": [1, 2, {"a": "b"}], "s": "x y z"} ' * 1000)
    d = b'["' + body.replace(b'"', b"'") * 1000 + b'", 1, 2]'
    ctx.set_tile_steps(1)
    try:
        _check(ctx, d)
        rng = random.Random(11)
        d2 = bytearray(b'{"a":[1,2,3],"b":"c"} ' * 1_000_000)
        for _ in range(3000):
            d2[rng.randrange(len(d2))] = 0x22
        _check(ctx, bytes(d2))
    finally:
        ctx.set_tile_steps(0)
    _check(ctx, d)


def test_fast_mode_timeout_falls_back_to_ticket_mode(twitter):
    """If a fast-mode launch reports a look-back timeout, the host path must re-run it in ticket mode, return
    correct results and stay in ticket mode."""
    import simdjson_java_amd as S
    c = S.Context(device=0, capacity=len(twitter) + 64)
    try:
        c.debug_set_flags(16)  # fake the timeout report of fast-mode launches
        idx, st = c.stage1(twitter)
        want, wst = O.stage1(twitter)
        assert st == wst == 0 and np.array_equal(idx, want)
        c.debug_set_flags(0)
        idx, st = c.stage1(twitter)   # now latched in ticket mode
        assert np.array_equal(idx, want)
    finally:
        c.close()


def test_few_resident_workgroups(twitter):
    """Far fewer granules in flight (8 worker workgroups = 64) than one scanner window holds (256): the scanner has to
    publish prefixes for the front part of a window whose back part has not been handed out yet, in both granule
    sizes, and the safe mode has to cope as well."""
    import simdjson_java_amd as S
    doc = twitter * 8  # 5 MB: 16 KiB granules
    want, wst = O.stage1(doc)
    for mode in (0, 1):
        c = S.Context(device=0, capacity=len(doc) + 64)
        try:
            c.set_tile_mode(mode)
            c.debug_set_flags(32)
            for steps in (0, 1, 2):
                c.set_tile_steps(steps)
                idx, st = c.stage1(doc)
                assert st == wst == 0 and np.array_equal(idx, want), (mode, steps)
        finally:
            c.close()


def test_index_capacity_error(ctx):
    import simdjson_java_amd as S
    with pytest.raises(S.SjmiError):
        ctx.stage1(b"[1,2,3,4,5,6,7,8]", index_capacity=4)


def test_device_path_twitter_x64(ctx, twitter):
    """Device-resident path: twitter.json x 64 (40 MB); closed form index[k*S+j] = k*N + index0[j]
    (scaled)."""
    import torch
    reps = 64
    n0 = len(twitter)
    idx0, st0 = O.stage1(twitter)
    assert st0 == 0
    host = torch.frombuffer(bytearray(twitter), dtype=torch.uint8)
    buf = torch.zeros(n0 * reps + 128, dtype=torch.uint8, device="cuda")
    buf[:n0 * reps] = host.cuda().repeat(reps)
    cap = idx0.size * reps + 1
    out = torch.empty(cap, dtype=torch.int32, device="cuda")
    res = torch.zeros(2, dtype=torch.int64, device="cuda")
    ctx.stage1_device(buf.data_ptr(), n0 * reps, out.data_ptr(), cap, res.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    r = res.cpu().numpy()
    assert int(r[0]) == idx0.size * reps and (int(r[1]) & 0xFFFFFFFF) == 0
    want = (torch.from_numpy(idx0.astype(np.int64)).cuda()[None, :] + (torch.arange(reps, device="cuda") * n0)[:, None]).flatten()
    got = out[:idx0.size * reps].to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(got, want)
    assert int(out[idx0.size * reps].item()) == 0


@pytest.mark.parametrize("unit,what", [(b'"abcdef",', "two fast rounds of two steps (1 structural per 4.5 bytes: 3,640 per 16 KiB granule)"),
                                       (b'12345,', "the windowed path (1 per 3 bytes: neither half of a granule fits the 2,304 staging slots)"),
                                       (b'"abcdefghijklmnopqr",', "one round (1 per 10.5 bytes)"),
                                       (b'"abcdef",' * 700 + b'12345,' * 900 + b'"abcdefghijklmnopqr",' * 300, "all three forms, changing inside granules")])
def test_index_emission_rounds_by_structural_density(ctx, unit, what):
    """k_stage1's expansion stages a 16 KiB granule's indexes in 2,304 LDS slots per wave: one round when they fit, two rounds of
    two steps when each half fits, windows otherwise (stage1.hip, fast_round): every form against the oracle, S = 4."""
    ctx.set_tile_steps(4)
    try:
        for total in (3 * 16384 + 100, 41 * 16384 + 5000):
            d = b"[" + unit * (total // len(unit)) + b"0]"
            _check(ctx, d)
    finally:
        ctx.set_tile_steps(0)


def _block_of_population(rng, k):
    """64 bytes with exactly k structurals, placed at random: operators for the structurals, spaces for the rest (every operator is a
    structural; nothing else here is)"""
    b = bytearray(b" " * 64)
    for p in rng.sample(range(64), k):
        b[p] = rng.choice(b"[]{}:,")
    return bytes(b)


@pytest.mark.parametrize("steps", [1, 2, 4])
def test_expansion_sorted_by_population(ctx, steps):
    """Round 6: the expansion hands a round's 32-bit half masks out again by population (stage1.hip sorted_round: counting sort in
    LDS, the rank of a half mask from an LDS atomic).  Blocks of EVERY population 0 .. 64 in shuffled order -- all buckets of the
    histogram in use inside one granule, empty half masks among full ones, rounds on both sides of the sort's threshold, granules
    that need one round / two rounds / the windowed path -- and runs of equal blocks (every lane in one bucket: the atomics of a
    whole wave on one counter)."""
    rng = random.Random(600 + steps)
    ctx.set_tile_steps(steps)
    try:
        for trial in range(6):
            blocks = []
            for _ in range(rng.choice([70, 300, 1100])):
                mode = rng.random()
                if mode < 0.5:
                    blocks.append(_block_of_population(rng, rng.randrange(65)))
                elif mode < 0.7:
                    blocks += [_block_of_population(rng, rng.choice([0, 1, 2, 31, 32, 33, 63, 64]))] * rng.randrange(1, 70)
                elif mode < 0.85:
                    blocks += [b" " * 64] * rng.randrange(1, 130)   # sparse stretches: rounds below the threshold
                else:
                    k = rng.randrange(65)
                    blocks += [_block_of_population(rng, k) for _ in range(rng.randrange(1, 64))]
            d = b"".join(blocks)
            d = d[:len(d) - rng.randrange(64)]  # (a ragged tail block)
            _check(ctx, d)
    finally:
        ctx.set_tile_steps(0)
