"""GPU stage 2 for batches (sjmi_walk_batch_device, csrc/walk.hip): one lane per document walks the structurals and
builds the tape.  Parity: tape words, string offsets and per-document error codes equal the host walker's (which equal
the oracle's); documents the GPU hands back to the host (deep nesting, floats outside the exact range) are exactly
those, and nothing else."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import load_fixture
from tests.test_gpu_batch import _pack, _small_docs
from tests.walk_common import NEEDS_HOST, number_documents

pytestmark = pytest.mark.gpu



def gpu_walk(ctx, docs, max_depth=1024):
    """isolated stage 1 -> strings -> walk, all device-resident; -> (tapes list / None, strings bytes, errors int32)."""
    import torch
    buf, offs = _pack(docs)
    n = len(docs)
    d_buf = torch.zeros(len(buf) + 128, dtype=torch.uint8, device="cuda")
    d_buf[:len(buf)] = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    d_idx = torch.zeros(len(buf) + 2, dtype=torch.int32, device="cuda")
    d_io = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    d_ds = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
    d_res = torch.zeros(2, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ctx.stage1_batch_isolated_device(d_buf.data_ptr(), len(buf), d_offs.data_ptr(), n, d_idx.data_ptr(), d_idx.numel(),
                                     d_io.data_ptr(), d_ds.data_ptr(), d_res.data_ptr(), stream)
    torch.cuda.synchronize()
    count = int(d_res[0].item())
    d_sb = torch.zeros(len(buf) + 4 * count + 64, dtype=torch.uint8, device="cuda")
    d_dso = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    d_ures = torch.zeros(3, dtype=torch.int64, device="cuda")
    ctx.unescape_batch_device(d_buf.data_ptr(), len(buf), d_idx.data_ptr(), count, d_offs.data_ptr(), d_io.data_ptr(), n,
                              d_sb.data_ptr(), d_sb.numel(), d_dso.data_ptr(), d_ures.data_ptr(), stream)
    cap = 2 * count + 2 * n + 8
    d_tape = torch.zeros(cap, dtype=torch.int64, device="cuda")
    d_to = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    d_err = torch.zeros(max(n, 1), dtype=torch.int32, device="cuda")
    d_wres = torch.zeros(4, dtype=torch.int64, device="cuda")
    ctx.walk_batch_device(d_buf.data_ptr(), d_offs.data_ptr(), n, d_idx.data_ptr(), count, d_io.data_ptr(), d_ds.data_ptr(),
                          d_sb.data_ptr(), d_dso.data_ptr(), 0, max_depth, d_tape.data_ptr(), cap, d_to.data_ptr(),
                          d_err.data_ptr(), d_wres.data_ptr(), stream)
    torch.cuda.synchronize()
    wres = d_wres.cpu().numpy()
    to = d_to.cpu().numpy().astype(np.int64)
    errors = d_err.cpu().numpy()[:n]
    assert int(wres[0]) == int(to[-1]) and (int(wres[3]) & 1) == 0
    assert int(wres[1]) == int((errors == NEEDS_HOST).sum()) and int(wres[2]) == int((errors > 0).sum())
    tape = d_tape.cpu().numpy().view(np.uint64)
    total = int(d_ures[0].item())
    strings = bytes(d_sb[:total].cpu().numpy())
    tapes = [tape[to[k]:to[k + 1]] if errors[k] == 0 else None for k in range(n)]
    for k in range(n):
        if errors[k] != 0:
            assert to[k + 1] == to[k]
    return tapes, strings, errors


def check_against_oracle(docs, tapes, strings, errors, host_ok=()):
    for k, d in enumerate(docs):
        want = O.parse(d + b"\n")
        if int(errors[k]) == NEEDS_HOST:
            assert k in host_ok, (k, d[:60])
            continue
        assert k not in host_ok, (k, d[:60])
        assert int(errors[k]) == want.error, (k, d[:60], int(errors[k]), want.error)
        if not want.error:
            assert tapes[k].size == want.tape.size, (k, d[:60])  # (STRING payloads differ: batch-wide string buffer)
            assert O.Parsed(tapes[k], strings, 0, 0, 0).to_python() == want.to_python(), k


@pytest.fixture(scope="module")
def ctx():
    import simdjson_java_amd as S
    c = S.Context(device=0, capacity=8 * 1024 * 1024)
    yield c
    c.close()


def test_walk_small_documents_and_errors(ctx):
    rng = random.Random(90)
    docs = _small_docs(rng, 4000)
    bad = [b"[1 1]", b"[1,,1]", b'{"a" 1}', b"[1,2", b'{"a":1,}', b"tru", b"[01]", b'["\\q"]', b'["\\uD800"]', b"1 2", b"[-]",
           b'["abc', b'{"k": "v', bytes([0x5B, 0x22, 0xC3, 0x22, 0x5D]), b'["a\x01b"]', b"", b"{", b"}", b"[}", b"{]", b'{"a":}',
           b'{1:2}', b'[1,]', b'{"a":1 "b":2}', b'[[[]]', b'[]]', b'{"a":{}', b'"', b'nul', b'nulll', b'falsee', b'[tru]',
           b'[nul]', b'[fals]', b'[truex]', b'[+1]', b'[.5]', b'[1.]', b'[1.e3]', b'[1e]', b'[1e+]', b'[-]', b'[--1]', b'[1a]',
           b'[9223372036854775808]', b'[-9223372036854775809]', b'[12345678901234567890123]', b"[1] x", b"{} {}"]
    good = [b"true", b"false", b"null", b"0", b"-0", b"-0.0", b"1", b"-1", b"1.5", b"1e3", b"1E3", b"1e+3", b"1e-3", b"123.456e-2",
            b'"root string \\n"', b'""', b"[]", b"{}", b"[[]]", b"[{}]", b'{"a":[]}', b'{"a":{}}', b"[9223372036854775807]",
            b"[-9223372036854775808]", b"[0.1, 0.2, 0.3, 1e22, 1e-22, 9007199254740992.0, 9007199254740992e0, 1234567890.123456]",
            b"[0.000000000000000000001]", b"[100000000000000000000.0]", b"[1.0000000000000000000]", b"[0e0, 0.0e5, -0e-3]",
            b" [1, 2] ", b"\t{\"k\" : [true, false, null]}\r", load_fixture("github_events.json").rstrip()]
    for b in bad + good:
        docs.insert(rng.randrange(len(docs)), b)
    tapes, strings, errors = gpu_walk(ctx, docs)
    assert not (errors == NEEDS_HOST).any()
    check_against_oracle(docs, tapes, strings, errors)
    assert int((errors > 0).sum()) >= len(bad) - 2


def test_walk_decides_hard_documents_on_the_device(ctx):
    """What rounds 1-2 handed back to the host walker is produced by the HIP path: floats of more than 19 significant digits
    within 10^-19 of a rounding boundary (the walker lists them, k_slow_doubles compares them exactly with the midpoint of
    the two candidates: csrc/sj_bigdec.h, DoubleParser.java:216-330) and nesting beyond the 64 levels in registers (levels
    64 .. 1023 in global memory).  No document of this batch may come back as SJMI_WALK_NEEDS_HOST; the reference's own depth
    limit still wins when it is lower."""
    from tests.walk_common import AMBIGUOUS
    hard = [("[%s]" % a).encode() for a in AMBIGUOUS[:5]] + [b"[" * 65 + b"]" * 65, b"[" * 64 + b"1" + b"]" * 64, b"[" * 100 + b"1" + b"]" * 100,
                                                             AMBIGUOUS[5].encode(), ("[-%s]" % AMBIGUOUS[6]).encode()]
    easy = [b"[123456789012345678901.5]", b"[12345678901234567891e0]", b"[1.2345678901234567890123]", b"[0.10000000000000000000000000000000000001]",
            b"3.141592653589793238462643383279", b"[100000000000000000000000000000.0000000000000000000001e-10]", b"[0.00000000000000000000000000012345678901234567890123456789]",
            b"[12345678901234567890123456789012345678901234567890e300]", b"[1.00000000000000000000000000000000000000000000001e-330]",
            b"[1e22]", b"[" * 64 + b"]" * 64, b"[" * 63 + b"1" + b"]" * 63, b"[9007199254740992.0]", b"[1.5]", b'{"a": [1, {"b": 2.25}]}',
            b"[1e23]", b"[1e-23]", b"[0.1e400]", b"[9007199254740993.0]", b"[1.7976931348623157e308]", b"[4.9e-324]", b"[2.4e-324]",
            b"[2.2250738585072013e-308]", b"[-1e999]", b"[1e-999]", b"[123456789012345678e0]", b"[12345678901234567890e0]"]
    docs = []
    for h, e in zip(hard, easy):
        docs.append(e)
        docs.append(h)
    docs += easy
    tapes, strings, errors = gpu_walk(ctx, docs)
    assert not (errors == NEEDS_HOST).any()
    check_against_oracle(docs, tapes, strings, errors)
    # maxDepth below the device stack: the reference's depth error, not a hand-back
    deep = [b"[" * 10 + b"]" * 10, b"[[1]]", b"[" * 9 + b"]" * 9]
    tapes, strings, errors = gpu_walk(ctx, deep, max_depth=10)
    for k, d in enumerate(deep):
        want = O.parse(d + b"\n", max_depth=10)
        assert int(errors[k]) == want.error, (k, int(errors[k]), want.error)
        if not want.error:
            assert np.array_equal(tapes[k], want.tape)


def test_walk_large_batch_against_the_oracle():
    """30,000 documents (300 of them malformed): the GPU walker's tapes, string records and error codes against the ORACLE,
    document by document -- and the host walker (SimdJsonParser.parse_batch, the C++ mirror of JsonIterator / TapeBuilder) against
    the oracle on the same batch, so both placements of stage 2 are pinned to the same checker rather than to each other."""
    import simdjson_java_amd as S
    import os
    rng = random.Random(91)
    docs = _small_docs(rng, 30000)
    for i in range(300):
        docs.insert(rng.randrange(len(docs)), [b"[1 1]", b'["abc', b"", b"[-]", b'"', b"nul", b'{"a":1,}'][i % 7])
    buf, offs = _pack(docs)
    os.environ["SJMI_PARSE_PIPELINE"] = "1"
    try:
        p = S.SimdJsonParser(capacity=len(buf) + 64)
    finally:
        del os.environ["SJMI_PARSE_PIPELINE"]
    c = S.Context(device=0, capacity=len(buf) + 64)
    try:
        tapes, strings, errors = gpu_walk(c, docs)
        check_against_oracle(docs, tapes, strings, errors)
        assert int((errors > 0).sum()) == 300
        htapes, hstrings, herrors = p.parse_batch(buf, offs)
        check_against_oracle(docs, htapes, hstrings, herrors)
    finally:
        p.close()
        c.close()


def test_walk_number_fuzz(ctx):
    """Random number literals in arrays: every value the GPU converts (Clinger / Eisel-Lemire, csrc/sj_number.h; literals of
    more than 19 significant digits whose two 19-digit neighbours round differently -- generated around exact midpoints --
    by the exact comparison of csrc/sj_bigdec.h in k_slow_doubles) equals the oracle's (strtod, correctly rounded) bit for
    bit.  Nothing is handed back."""
    rng = random.Random(92)
    docs, hard, either = number_documents(rng, 6000)
    tapes, strings, errors = gpu_walk(ctx, docs)
    assert 20 < len(hard) < 1000  # (the generator does aim at the boundaries)
    for k, d in enumerate(docs):
        want = O.parse(d + b"\n")
        assert int(errors[k]) == want.error == 0, (k, d, int(errors[k]), want.error)
        assert np.array_equal(tapes[k], want.tape), (k, d, k in hard)


def test_reference_number_vectors_on_the_gpu(ctx):
    """All 158 literal inputs of NumberParsingTest.java through the GPU walk as one batch: the asserted bits / long /
    error ON THE DEVICE for every one of them: at most 19 significant digits (Eisel-Lemire: ties to even, round up / down,
    subnormal and normal boundaries, +-infinity, signed zeros, exponents longer than a long), longer ones that their two
    19-digit neighbours decide, and the two exact midpoints with a tail behind the 19th digit (k_slow_doubles)."""
    from tests.conftest import number_vectors
    vs = [v for v in number_vectors()]
    docs = [v["input"].encode("utf-8")[:v.get("length")] for v in vs]
    tapes, strings, errors = gpu_walk(ctx, docs)
    on_device = 0
    for k, v in enumerate(vs):
        want = O.parse(docs[k] + b"\n")
        assert int(errors[k]) != NEEDS_HOST, v["input"][:60]
        assert int(errors[k]) == want.error, (v["input"][:40], int(errors[k]), want.error)
        if "message" in v:
            assert want.error != 0 and O.error_message(want.error) == v["message"], v["input"][:40]
        else:
            got = O.Parsed(tapes[k], strings, 0, 0, 0).to_python()
            assert got == (("l", v["long"]) if "long" in v else ("d", v["double_bits"])), (v["input"][:40], v["cite"], got)
        on_device += 1
    assert on_device == len(vs) >= 158


def test_outputs_do_not_depend_on_the_run(ctx):
    """The same batch of token-level adversarial documents 25 times through the three calls: every tape, the string buffer and every
    verdict identical to the first run's (which is checked against the oracle) -- nothing in the kernels may depend on timing
    (tools/determinism_stress.py is the long form: 500 k repeated runs)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from token_docs import document
    sys.setrecursionlimit(20000)
    rng = random.Random(2606)
    docs = [document(rng) for _ in range(1500)]
    tapes0, strings0, errors0 = gpu_walk(ctx, docs)
    check_against_oracle(docs, tapes0, strings0, errors0)
    for _ in range(25):
        tapes, strings, errors = gpu_walk(ctx, docs)
        assert np.array_equal(errors, errors0) and strings == strings0
        assert all((a is None and b is None) or np.array_equal(a, b) for a, b in zip(tapes, tapes0))
