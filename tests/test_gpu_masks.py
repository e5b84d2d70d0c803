"""Bit-mask parity (north_star: "bit-exact with the reference's own stage-1 bitmasks"): sjmi_stage1_masks on the GPU
against the oracle's line-by-line restatement of StructuralIndexer.java:210-252, six 64-bit masks per 64-byte block
{escaped, quote, inString, op, whitespace, structurals}; and the structurals mask against the indexes the streaming
kernel (k_stage1) emits for the same bytes, so that the two device formulations are tied together as well."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import load_fixture
from tests.golden import vectors as V

pytestmark = pytest.mark.gpu

MASK_NAMES = ["escaped", "quote", "inString", "op", "whitespace", "structurals"]


@pytest.fixture(scope="module")
def ctx():
    import simdjson_java_amd as S
    c = S.Context(device=0, capacity=48 * 1024 * 1024)
    yield c
    c.close()


def _check(ctx, d, indexes_too=True):
    want_idx, want_st, want = O.index_blocks(d, want_masks=True)
    got = ctx.stage1_masks(d)
    assert got.shape == want.shape
    if not np.array_equal(got, want):
        b, k = [int(x[0]) for x in np.nonzero(got != want)]
        raise AssertionError("block %d mask %s: got %016x want %016x" % (b, MASK_NAMES[k], int(got[b, k]), int(want[b, k])))
    if indexes_too:
        idx, _st = ctx.stage1(d)
        bits = np.unpackbits(got[:, 5].copy().view(np.uint8), bitorder="little")
        assert np.array_equal(np.nonzero(bits)[0].astype(np.uint32), idx)


@pytest.mark.parametrize("name", ["twitter.json", "github_events.json", "wide_bench.json", "malformed.txt"])
def test_reference_files(ctx, name):
    _check(ctx, load_fixture(name))


def test_reference_structural_indexer_inputs(ctx):
    for case in V.STRUCTURAL_INDEXER:  # StructuralIndexerTest.java:14-273
        _check(ctx, case[1])


def test_fuzz(ctx):
    rng = random.Random(31)
    alphabet = b'\\\\\\"""{}[]:, \t\n\r\x0c\x1a\x01abc019.-e\xc3\xa9'
    for it in range(400):
        n = rng.choice([0, 1, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 16384, 16385, rng.randint(0, 40000)])
        mode = it % 4
        if mode == 0:
            d = bytes(rng.choice(alphabet) for _ in range(n))
        elif mode == 1:
            d = bytes(rng.choice(b'\\"a ') for _ in range(n))
        elif mode == 2:
            d = b"a" * rng.randint(0, 70) + b"\\" * rng.randint(1, 300) + rng.choice([b'"', b"x", b""]) + b'"x' * rng.randint(0, 40)
        else:
            d = bytes(rng.getrandbits(8) for _ in range(n))
        _check(ctx, d)


def test_long_string_and_backslash_runs(ctx):
    """inString = all-ones for megabytes (the parity has to come through the word scan), a backslash run longer than any
    halo, lone quotes sprinkled over a large document."""
    body = b'{"k": [1, 2, {"a": "b"}], "s": "x y z"} ' * 20000
    _check(ctx, b'["' + body.replace(b'"', b"'") + b'", 1, 2]')
    _check(ctx, b"\\" * (1 << 18) + b'"x"')
    rng = random.Random(32)
    d2 = bytearray(b'{"a":[1,2,3],"b":"c"} ' * 100000)
    for _ in range(500):
        d2[rng.randrange(len(d2))] = 0x22
    _check(ctx, bytes(d2))


def test_twitter_x64_device_resident(ctx, twitter):
    """Device-resident form on 40 MB: block b of copy k is not aligned with block b of copy 0 (631,515 is odd), so every
    copy exercises different block phases; compared with the oracle on the whole concatenation."""
    import torch
    reps = 64
    doc = twitter * reps
    n = len(doc)
    nb = n // 64 + 1
    buf = torch.zeros(n + 128, dtype=torch.uint8, device="cuda")
    buf[:n] = torch.frombuffer(bytearray(doc), dtype=torch.uint8).cuda()
    masks = torch.zeros(nb * 6, dtype=torch.int64, device="cuda")
    ctx.stage1_masks_device(buf.data_ptr(), n, masks.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = masks.cpu().numpy().view(np.uint64).reshape(nb, 6)
    _, _, want = O.index_blocks(doc, want_masks=True)
    assert np.array_equal(got, want)
