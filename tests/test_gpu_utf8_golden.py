"""The reference's own UTF-8 verdict vectors (Utf8ValidationTest.java:37-424, testutils/Utf8TestData.java:18-33;
transcribed in tests/golden/vectors.py) through the HIP kernels, via the C ABI.

The reference embeds every invalid sequence in random valid UTF-8 (unseeded); here the surrounding text is seeded
and the sequence is PLACED so that it straddles exactly the boundaries that exist only in the GPU formulation:
  * the 64-byte block of a lane (halo carry from the 8 bytes before the block, sj_block.h sj_utf8_carry),
  * the 4 KiB wave-step (64 lanes; the per-wave "all ASCII" ballot that skips the UTF-8 algebra, stage1.hip),
  * the chain granule (4 KiB x steps: another wave's registers),
  * the 1 KiB row step of the isolated-batch kernel (16 lanes per document, batch.hip k_doc_pass),
  * the very end of the document (tail masking, sj_mask_tail; Utf8Validator.java:115-117,165).
Expectation per document = the oracle's status for the same bytes (pinned to "invalid" for these families by
tests/test_oracle_golden.py); the UTF-8 bit must be set for every one of them."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.golden import vectors as V

pytestmark = pytest.mark.gpu

_CHARS = ["a", "b", "z", "0", "7", " ", "e", "\u00e9", "\u00fc", "\u07ff", "\u0800", "\u20ac", "\ud55c", "\uffff",
          "\U00010000", "\U0001f600", "\U0010ffff"]
_ENC = [c.encode("utf-8") for c in _CHARS]


def _filler(rng, nbytes):
    """Exactly nbytes of valid UTF-8 (no quotes / backslashes / control characters), ending on a character boundary."""
    out = bytearray()
    while len(out) < nbytes:
        c = rng.choice(_ENC)
        if len(out) + len(c) > nbytes:
            c = b"x"
        out += c
    return bytes(out)


def _embed(rng, seq, boundary, split, tail):
    """seq placed so that `split` of its bytes lie in front of byte offset `boundary`; `tail` bytes of valid text after it."""
    assert boundary >= split
    return _filler(rng, boundary - split) + seq + _filler(rng, tail)


def _all_sequences():
    """(name, seq, at_end_only) for every vector of the three groups."""
    out = []
    for name, seq, _cite in V.UTF8_INVALID_MID:
        out.append((name, seq, False))
    for name, seq, _cite in V.UTF8_INVALID_END:
        out.append((name, seq, True))
    for name, seqs, _cite in V.UTF8_INVALID_FAMILIES:
        for s in seqs:
            out.append((name, s, False))
    return out


@pytest.fixture(scope="module", params=["fast", "ticket"])
def ctx(request):
    import simdjson_java_amd as S
    c = S.Context(device=0, capacity=1 << 20)
    c.set_tile_mode(request.param == "ticket")
    c.mode = request.param
    yield c
    c.close()


def _expect_invalid(ctx, d, what):
    want_idx, want_st = O.stage1(d)
    assert want_st & O.ST_UTF8, ("oracle accepts", what, d[-16:].hex())
    got_idx, got_st = ctx.stage1(d)
    assert got_st == want_st, (what, got_st, want_st)
    assert np.array_equal(got_idx, want_idx), what  # indexes are written whatever the verdict


@pytest.mark.parametrize("steps", [0, 4])
def test_named_sequences_at_every_boundary_and_split(ctx, steps):
    """Utf8ValidationTest.java:71-329: the 9 mid-text and 6 at-the-end sequences, every split position across block,
    wave-step and granule boundaries (granule = 4 KiB with automatic steps on these sizes, 16 KiB with steps = 4)."""
    ctx.set_tile_steps(steps)
    try:
        rng = random.Random(1000 + steps)
        g = 4096 * (steps or 1)
        bounds = [64, 128, 4096 - 64, 4096, 4096 + 64, g, 2 * g, g + 4096]
        for name, seq, _cite in V.UTF8_INVALID_MID:
            for b in bounds:
                for split in range(len(seq) + 1):
                    _expect_invalid(ctx, _embed(rng, seq, b, split, rng.choice([1, 5, 70, 300])), (name, b, split))
        for name, seq, _cite in V.UTF8_INVALID_END:
            for b in bounds:
                for split in range(len(seq) + 1):  # document ends right behind the truncated sequence
                    _expect_invalid(ctx, _embed(rng, seq, b, split, 0), (name, b, split))
            for n in (1, 2, 3, 63, 64, 65, 4095, 4096, 4097):  # and at every alignment of the end itself
                if n >= len(seq):
                    _expect_invalid(ctx, _filler(rng, n - len(seq)) + seq, (name, "len", n))
        # controls: VALID characters straddling the same boundaries must not trip anything
        for c in ("\u00e9", "\u20ac", "\U0001f600", "\U0010ffff", "\ud7ff", "\ue000"):
            e = c.encode()
            for b in bounds:
                for split in range(len(e) + 1):
                    d = _embed(rng, e, b, split, rng.choice([0, 3, 100]))
                    idx, st = ctx.stage1(d)
                    widx, wst = O.stage1(d)
                    assert st == wst == 0 and np.array_equal(idx, widx), (c, b, split)
    finally:
        ctx.set_tile_steps(0)


def test_families_through_the_streaming_kernel(ctx):
    """Utf8ValidationTest.java:37-69,331-424: EVERY sequence of every family in FAST mode (71,000+ documents; in SAFE
    mode every sequence of the small families and every 5th of the large ones -- overlong 3/4-byte, surrogates, generated
    by Utf8TestData.utf8Sequences), the boundary and split rotating with the sequence number so that each family meets
    each boundary kind in each split."""
    rng = random.Random(2000)
    bounds = [64, 4096, 8192, 192, 4096 + 128]
    k = 0
    for name, seqs, _cite in V.UTF8_INVALID_FAMILIES:
        stride = 1 if len(seqs) <= 256 or ctx.mode == "fast" else 5
        for j in range(0, len(seqs), stride):
            seq = seqs[(j + (k % stride)) % len(seqs)] if stride > 1 else seqs[j]
            b = bounds[k % len(bounds)]
            split = (k // len(bounds)) % (len(seq) + 1)
            tail = (0, 2, 90)[(k // 7) % 3]
            _expect_invalid(ctx, _embed(rng, seq, b, split, tail), (name, j, b, split))
            k += 1
    assert k > (70000 if ctx.mode == "fast" else 14000)


def test_every_sequence_as_a_document_of_an_isolated_batch():
    """ALL 71,000+ sequences of the three groups, each in its own document of ONE isolated batch (batch.hip: one DPP row of
    16 lanes per document, 1 KiB per step): sequence at the document start, across a 64-byte lane boundary, across the
    1 KiB / 2 KiB row-step boundaries and at the very end of the document (no separator behind it: the next document's
    first byte must stay invisible).  Every document must get the oracle's status, the valid control documents mixed in
    must stay clean with their own indexes."""
    import simdjson_java_amd as S
    rng = random.Random(3000)
    docs = []
    kinds = 0
    for i, (name, seq, at_end) in enumerate(_all_sequences()):
        for rep in range(2):
            kind = (2 * i + rep) % 6
            kinds |= 1 << kind
            split = (i // 3 + rep) % (len(seq) + 1)
            if at_end or kind == 0:
                b = (64, 1024, 70, 2048, 1, 130)[(i + rep) % 6]
                b = max(b, len(seq))
                docs.append(_filler(rng, b - len(seq)) + seq if kind % 2 else _embed(rng, seq, b, min(split, b), 0))
            elif kind == 1:
                docs.append(seq + _filler(rng, rng.randint(0, 80)))
            elif kind == 2:
                docs.append(_embed(rng, seq, 64, split, rng.randint(1, 40)))
            elif kind == 3:
                docs.append(_embed(rng, seq, 1024, split, rng.randint(1, 40)))
            elif kind == 4:
                docs.append(_embed(rng, seq, 2048, split, rng.randint(0, 3)))
            else:
                docs.append(_embed(rng, seq, 128 + 64 * (i % 13), split, rng.randint(1, 200)))
        if i % 50 == 0:  # valid neighbours
            docs.append(_filler(rng, rng.choice([0, 1, 63, 64, 65, 1023, 1024, 1025, 2050])))
            docs.append(b'["' + _filler(rng, rng.randint(0, 1100)) + b'", 1]')
    assert kinds == 0x3F
    offs = np.zeros(len(docs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])
    buf = b"".join(docs)
    ctx = S.Context(0, len(buf) + 64)
    try:
        idx, io, ds, st = ctx.stage1_batch_isolated(buf, offs)
        n_bad = 0
        for k, d in enumerate(docs):
            want, wst = O.stage1(d)
            assert int(ds[k]) == wst, (k, d[-24:].hex(), int(ds[k]), wst)
            got = idx[int(io[k]):int(io[k + 1])].astype(np.int64) - int(offs[k])
            if wst:
                n_bad += 1
                assert got.size == 0
            else:
                assert np.array_equal(got, want.astype(np.int64)), k
        assert n_bad >= 2 * len(_all_sequences())
        assert st & O.ST_UTF8
    finally:
        ctx.close()
