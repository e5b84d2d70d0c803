"""Zero-copy outputs of the host entry points (include/sjmi.h, sjmi_host_register): with page-locked, device-visible output
arrays the kernels write indexes / string records / tape straight into the caller's memory.  The bytes must be the ones the
staged path (pageable arrays: device buffers + download) delivers, for accepted and for rejected documents, through
registration changes, and for the two-call forms that read the indexes of the previous call back on the device."""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import load_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import simdjson_java_amd as S
    c = S.Context(device=0, capacity=4 * 1024 * 1024)
    yield c
    c.close()


def _pinned(n, dtype):
    import torch
    t = torch.empty(n, dtype={np.uint32: torch.int32, np.uint8: torch.uint8}[dtype]).pin_memory()
    return t, t.numpy().view(dtype)


DOCS = [
    b'{"a":[1,2,{"b":"x\\ny"}],"c":"\\u00e9\\ud83d\\ude00","d":null}',
    b'[]',
    b'',
    b'"lonely \\t string"',
    b'{"bad":"\\q"}',                  # string error
    b'{"a":"\xff"}',                   # UTF-8 error: stage 1 rejects
    b'{"unclosed":"abc',               # unclosed string
    b'[' + b','.join(b'"s%d\\\\"' % i for i in range(3000)) + b']',
]


def _staged(ctx, doc):
    idx, st, sb, fei, fec = ctx.stage1_unescape(doc)
    return idx.copy(), st, bytes(sb), fei, fec


@pytest.mark.parametrize("k", range(len(DOCS)))
def test_fused_call_with_pinned_outputs_equals_the_staged_call(ctx, k):
    doc = DOCS[k]
    want = _staged(ctx, doc)
    n = len(doc)
    keep_i, idx = _pinned(n + 2 + 64, np.uint32)
    keep_s, sb = _pinned(n + 4 * (n // 2 + 2) + 64, np.uint8)
    idx[:] = 0xDEADBEEF
    sb[:] = 0xAB
    got = ctx.stage1_unescape(doc, idx=idx, sb=sb)
    assert (got[1], got[3], got[4]) == (want[1], want[3], want[4])
    assert np.array_equal(got[0], want[0])
    if want[1] == 0 and want[3] is None:
        assert bytes(got[2]) == want[2]
    if want[1] == 0:
        assert idx[want[0].size] == 0, "sentinel"


def test_reference_files_and_the_oracle(ctx):
    for name in ("twitter.json", "github_events.json"):
        doc = load_fixture(name)
        n = len(doc)
        keep_i, idx = _pinned(n + 2, np.uint32)
        keep_s, sb = _pinned(n + 4 * (n // 2 + 2) + 64, np.uint8)
        for _ in range(3):  # (the cached view is used from the second call on)
            got_idx, st, got_sb, fei, fec = ctx.stage1_unescape(doc, idx=idx, sb=sb)
            want_idx, want_st = O.stage1(doc)
            assert st == want_st == 0 and fei is None
            assert np.array_equal(got_idx, want_idx)
            want_sb = O.unescape_all(doc + b"\0" * 64, want_idx)[0]
            assert bytes(got_sb) == want_sb


def test_string_capacity_too_small_is_reported_not_overrun(ctx):
    import simdjson_java_amd as S
    doc = b'[' + b','.join(b'"abcdefgh"' for _ in range(500)) + b']'
    keep_i, idx = _pinned(len(doc) + 2, np.uint32)
    keep_s, sb_full = _pinned(8192, np.uint8)
    sb_full[:] = 0x5A
    sb = sb_full[:1024]  # (capacity 1024 of a pinned page: 500 * 12 = 6000 bytes are needed)
    with pytest.raises(S.SjmiError) as e:
        ctx.stage1_unescape(doc, idx=idx, sb=sb)
    assert "capacity" in str(e.value)
    assert (sb_full[1024:] == 0x5A).all(), "wrote behind string_capacity"


def test_two_call_forms_read_the_zero_copy_indexes(ctx):
    """sjmi_stage1 into a pinned array, then sjmi_unescape of 'the last document'."""
    rng = random.Random(7)
    for _ in range(5):
        parts = ['{"k%d":["%s",%d,{"x":"\\u00%02x"}]}' % (i, "ab\\n" * rng.randint(0, 9), rng.randint(-5, 5), rng.randint(0x20, 0x7e)) for i in range(rng.randint(1, 200))]
        doc = ("[" + ",".join(parts) + "]").encode()
        want_idx, want_st = ctx.stage1(doc)
        want_sb = ctx.unescape(len(doc) * 3 + 64)
        keep, idx = _pinned(len(doc) + 2, np.uint32)
        got_idx, st = ctx.stage1(doc, idx=idx)
        assert st == want_st and np.array_equal(got_idx, want_idx)
        assert ctx.unescape(len(doc) * 3 + 64) == want_sb


def test_registration_changes_drop_the_cached_views(ctx):
    """register -> zero-copy; unregister -> the same address is pageable again and takes the staged path (a stale device view
    would fault the GPU); register again -> zero-copy again.  Same results each time."""
    from simdjson_java_amd.binding import lib
    doc = load_fixture("twitter.json")
    n = len(doc)
    idx = np.empty(n + 2, dtype=np.uint32)
    sb = np.empty(n + 4 * (n // 2 + 2) + 64, dtype=np.uint8)
    want = _staged(ctx, doc)
    for state in ("pageable", "registered", "pageable", "registered", "pageable"):
        if state == "registered":
            assert lib().sjmi_host_register(ctx._h, C.c_void_p(idx.ctypes.data), C.c_uint64(idx.nbytes)) == 0
            assert lib().sjmi_host_register(ctx._h, C.c_void_p(sb.ctypes.data), C.c_uint64(sb.nbytes)) == 0
        try:
            for _ in range(2):
                idx[:] = 0
                got = ctx.stage1_unescape(doc, idx=idx, sb=sb)
                assert np.array_equal(got[0], want[0]) and bytes(got[2]) == want[2] and got[1] == 0
        finally:
            if state == "registered":
                assert lib().sjmi_host_unregister(ctx._h, C.c_void_p(idx.ctypes.data)) == 0
                assert lib().sjmi_host_unregister(ctx._h, C.c_void_p(sb.ctypes.data)) == 0


def test_batch_after_a_zero_copy_call_reads_its_own_indexes(ctx):
    """a zero-copy sjmi_stage1 leaves 'the last call's indexes' in the CALLER's array; a later host batch call writes
    the context's own array -- the string pass of that batch must locate a failing string in the batch's indexes, not in the
    previous call's (the caller's array may even be unregistered by then)."""
    from simdjson_java_amd.binding import lib
    single = b'{"first":["call","with","quite","a","few","strings","so","that","the","arrays","differ"]}'
    idx_host = np.empty(len(single) + 2 + 64, dtype=np.uint32)
    assert lib().sjmi_host_register(ctx._h, C.c_void_p(idx_host.ctypes.data), C.c_uint64(idx_host.nbytes)) == 0
    try:
        got_idx, st = ctx.stage1(single, idx=idx_host)
        assert st == 0 and np.array_equal(got_idx, O.stage1(single)[0])
    finally:
        assert lib().sjmi_host_unregister(ctx._h, C.c_void_p(idx_host.ctypes.data)) == 0
    idx_host[:] = 0xFFFFFFFF  # (whatever a stale read would find)
    docs = [b'{"a":"fine"}\n', b'[1,2,3]\n', b'{"k":["x","y","bad \\q escape","z"]}\n', b'"tail"\n']
    buf = b"".join(docs)
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
    idx, io, ds, st = ctx.stage1_batch_isolated(buf, offs)
    assert st == 0 and not ds.any()
    want_idx = O.stage1(buf)[0]
    assert np.array_equal(idx, want_idx)
    sb, dso, fei, fec = ctx.unescape_batch(len(buf) * 2 + 64, len(docs))
    want_sb, _, want_feo, want_code = O.unescape_all(buf + b"\0" * 64, want_idx)
    quote_positions = [i for i in range(idx.size) if buf[idx[i]] == 0x22]  # oracle: ordinal among strings; GPU: position in indexes[]
    assert want_feo >= 0 and fei == quote_positions[want_feo]
    assert buf[idx[fei] + 1:idx[fei] + 5] == b"bad " and fec == want_code
    # and the single-document two-call form right behind an unregistration refuses instead of reading through a dead view
    import simdjson_java_amd as S
    assert lib().sjmi_host_register(ctx._h, C.c_void_p(idx_host.ctypes.data), C.c_uint64(idx_host.nbytes)) == 0
    ctx.stage1(single, idx=idx_host)
    assert lib().sjmi_host_unregister(ctx._h, C.c_void_p(idx_host.ctypes.data)) == 0
    with pytest.raises(S.SjmiError):
        ctx.unescape(4096)
