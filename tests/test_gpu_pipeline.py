"""sjmi_parse_batch_device (isolated stage 1 -> string records -> GPU walk queued without a host round trip; what one
rank of the sharded multi-GPU batch runs, simdjson-java_amd/sharding.py BatchShard): identical outputs to the three
separate calls, per-document parity with the oracle, and capacity shortfalls reported -- never overrun."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_gpu_batch import _pack, _small_docs
from tests.test_gpu_walk import gpu_walk

pytestmark = pytest.mark.gpu


def _docs():
    rng = random.Random(123)
    docs = _small_docs(rng, 5000)
    for i in range(60):
        docs.insert(rng.randrange(len(docs)), [b"[1 1]", b'["abc', b"", b"[-]", b'"', b"nul", b'{"a":1,}', b'["\\q"]',
                                               bytes([0x5B, 0x22, 0xC3, 0x22, 0x5D]), b'"root \\t string"'][i % 10])
    return docs


def test_fused_pipeline_equals_the_three_calls_and_the_oracle():
    import torch
    import simdjson_java_amd as S
    from simdjson_java_amd import sharding
    docs = _docs()
    buf, offs = _pack(docs)
    ctx = S.Context(0, 1 << 20)
    try:
        shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
        for _ in range(2):  # (twice: the second call reuses every workspace)
            shard.step(torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        c = shard.check()
        tapes, strings, errors = gpu_walk(ctx, docs)
        to = shard.tape_offsets.cpu().numpy()
        tape = shard.tape.cpu().numpy().view(np.uint64)
        err = shard.doc_errors.cpu().numpy()[:len(docs)]
        assert np.array_equal(err, errors)
        assert bytes(shard.sb[:c["string_bytes"]].cpu().numpy()) == strings
        n_bad = 0
        for k, d in enumerate(docs):
            got = tape[int(to[k]):int(to[k + 1])]
            if errors[k] != 0:
                # (a failing document has no tape.  Its slot holds unspecified words of its PREDICTED length when the tapes were
                #  laid out before the walk -- the accepted plain pass, and since round 6 the repair pass that takes this batch with
                #  its documents that fail stage 1: two words for those -- and is empty when the tapes are packed behind the walk)
                assert got.size == 0 or int(shard.doc_status.cpu().numpy()[k]) == 0 or got.size == 2
                n_bad += 1
                continue
            assert np.array_equal(got, tapes[k]), k
            want = O.parse(d + b"\n")
            assert want.error == 0 and O.Parsed(got, strings, 0, 0, 0).to_python() == want.to_python(), k
        assert c["failed_documents"] == n_bad >= 50 and c["documents"] == len(docs)
        g = sharding.sharded_step(shard, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert g.cpu().numpy().tolist() == [[len(docs), c["structurals"], c["string_bytes"], n_bad]]
    finally:
        ctx.close()


@pytest.mark.parametrize("short", ["indexes", "strings", "tape"])
def test_capacity_shortfall_is_reported_not_overrun(short):
    """Too small an index array / string buffer / tape: every stage behind the failing one must leave the incomplete
    arrays alone (no GPU fault) and check() must raise."""
    import torch
    import simdjson_java_amd as S
    from simdjson_java_amd import sharding
    docs = _docs()
    buf, offs = _pack(docs)
    ctx = S.Context(0, 1 << 20)
    try:
        kw = {"indexes": dict(index_ratio=40), "strings": dict(string_ratio=0.01, index_ratio=3), "tape": dict(tape_ratio=0.001)}[short]
        shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0), **kw)
        if short == "strings":
            shard.sb_capacity = 1000  # (the constructor adds 4 bytes per possible string: shrink it for real)
        if short == "tape":
            shard.tape_capacity = 100
        shard.step(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError):
            shard.check()
        # the device is still healthy: a properly sized shard on the same context runs
        ok = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
        ok.step(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert ok.check()["documents"] == len(docs)
    finally:
        ctx.close()


def _run_shard(ctx, buf, offs, n, exact=False, want_rejected=None):
    """exact=False: sjmi_parse_batch_device_optimistic, and check() makes the exact call when the record says SJMI_ST_REJECTED;
    exact=True: sjmi_parse_batch_device (everything queued)."""
    import torch
    from simdjson_java_amd import sharding
    shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
    for _ in range(2):
        shard.step(torch.cuda.current_stream().cuda_stream, exact=exact)
        torch.cuda.synchronize()
    if not exact:
        st1 = int(shard.result.cpu().numpy()[1]) & 0xFFFFFFFF
        if want_rejected is not None:
            assert bool(st1 & 0x800) == want_rejected, hex(st1)
    c = shard.check()
    assert not (c["stage1_status"] & 0x800)
    if want_rejected is not None and not exact:
        assert getattr(shard, "rejected_steps", 0) == (1 if want_rejected else 0)
    to = shard.tape_offsets.cpu().numpy()
    tape = shard.tape.cpu().numpy().view(np.uint64)
    err = shard.doc_errors.cpu().numpy()[:n]
    io = shard.index_offsets.cpu().numpy()[:n + 1]
    idx = shard.idx.cpu().numpy()[:c["structurals"] + 1]
    return c, tape, to, err, bytes(shard.sb[:c["string_bytes"]].cpu().numpy()), io, idx


@pytest.mark.parametrize("exact", [False, True], ids=["optimistic_entry", "exact_entry"])
@pytest.mark.parametrize("separator", [b"\n", b"\r\n", b"\t", b" ", b""], ids=["lf", "crlf", "tab", "space", "none"])
def test_optimistic_plain_pass_and_its_rejections(separator, exact, monkeypatch):
    """sjmi_parse_batch_device indexes a batch with ONE plain k_stage1 launch when every document ends in a control-character
    separator and the global verdict is clean; anything else (space / no separators, any broken document) falls to the
    per-document passes.  Either way the outputs equal those of the pipeline with the optimistic pass switched off
    (SJMI_BATCH_OPTIMISTIC=0 in a fresh process is not needed: the rejected cases ARE the per-document passes) and the
    oracle's, document by document -- including batches in which two documents leave a string open (which a plain pass
    without the separator rule would accept)."""
    import simdjson_java_amd as S
    rng = random.Random(99)
    good = [d for d in _small_docs(rng, 3000)]
    cases = {"all valid": good,
             "two unclosed strings": good[:700] + [b'["abc'] + good[700:1500] + [b'def"]'] + good[1500:],
             "one broken": good[:100] + [b"[1 1]"] + good[100:],
             "utf-8": good[:50] + [bytes([0x5B, 0x22, 0xC3, 0x22, 0x5D])] + good[50:],
             "empty documents": good[:10] + [b"", b""] + good[10:40]}
    ctx = S.Context(0, 1 << 20)
    try:
        for name, docs in cases.items():
            if separator == b"" and name == "empty documents":
                continue
            buf = b"".join(d + separator for d in docs)
            offs = np.concatenate([[0], np.cumsum([len(d) + len(separator) for d in docs])]).astype(np.uint64)
            # which batches the optimistic pipeline takes: control-character separators and every document through stage 1
            # (b"[1 1]" passes stage 1: a grammar error is the walker's business, not a rejection)
            accepted = separator in (b"\n", b"\r\n", b"\t") and name in ("all valid", "one broken", "empty documents")
            c, tape, to, err, strings, io, idx = _run_shard(ctx, buf, offs, len(docs), exact=exact, want_rejected=not accepted)
            n_bad = 0
            for k, d in enumerate(docs):
                # (a document is judged alone: what stands behind it in the batch must not matter -- with space / no
                #  separators a scalar at the end of one document would otherwise merge with the next)
                want = O.parse(d + separator) if separator.strip() == b"" and separator else O.parse(d)
                got = tape[int(to[k]):int(to[k + 1])]
                assert int(err[k]) == want.error, (name, k, d[:40], int(err[k]), want.error)
                if want.error:
                    n_bad += 1  # (no tape; the slot's size says nothing: see include/sjmi.h, sjmi_parse_batch_device)
                else:
                    assert O.Parsed(got, strings, 0, 0, 0).to_python() == want.to_python(), (name, k)
            assert c["failed_documents"] == n_bad, name
    finally:
        ctx.close()


@pytest.mark.parametrize("exact", [False, True], ids=["optimistic_entry", "exact_entry"])
def test_a_batch_of_one_document(exact):
    """A shard that holds ONE document (a rank's share of a tiny batch): the optimistic entry has no token-walker path for it and
    must say SJMI_ST_REJECTED -- not fail with a HIP error and a half-written record (round 5's advisor finding) -- so that
    check() makes the exact call; the exact entry serves it directly.  Valid, failing stage 2, failing stage 1."""
    import simdjson_java_amd as S
    ctx = S.Context(0, 1 << 20)
    try:
        for d in (b'{"a":[1,2.5,"x\\n"],"b":{"c":null}}', b"[1 1]", b'["abc', b"7"):
            buf = d + b"\n"
            offs = np.array([0, len(buf)], dtype=np.uint64)
            c, tape, to, err, strings, io, idx = _run_shard(ctx, buf, offs, 1, exact=exact, want_rejected=True)
            want = O.parse(buf)
            assert int(err[0]) == want.error, (d, int(err[0]), want.error)
            assert c["failed_documents"] == (1 if want.error else 0) and c["host_documents"] == 0
            if not want.error:
                assert O.Parsed(tape[int(to[0]):int(to[1])], strings, 0, 0, 0).to_python() == want.to_python(), d
    finally:
        ctx.close()


def test_accepted_batch_with_strings_that_are_not_structurals():
    """A quote directly behind a primitive (1"abc", true"x") opens a string for the string pass without being a structural
    (StructuralIndexer.java:243-248: a scalar start needs a non-scalar in front of it).  Such a document passes stage 1 and fails
    stage 2 -- but the documents BEHIND it in the same 64-byte block take their first string ordinal from a count of the strings
    opened in front of them, and that count must see the quote (found by tools/soak_pipeline.py, round 4: the accepted plain
    pass counted '"' structurals).  Every document against the oracle, through the fused call."""
    import simdjson_java_amd as S
    rng = random.Random(5)
    good = _small_docs(rng, 400)
    odd = [b'1"abc"', b'true"x"', b'[1"a","b"]', b'{"k":2"v"}', b'null"n""m"', b'-0"z"']
    docs = []
    for i in range(1200):
        docs.append(odd[i % len(odd)] if i % 3 == 0 else (good[i % len(good)] if i % 3 == 1 else [b'"s%d"' % i, b'["a","b%d"]' % i, b'{"q":"r"}'][i % 9 // 3]))
    ctx = S.Context(0, 1 << 20)
    try:
        buf = b"".join(d + b"\n" for d in docs)
        offs = np.concatenate([[0], np.cumsum([len(d) + 1 for d in docs])]).astype(np.uint64)
        c, tape, to, err, strings, io, idx = _run_shard(ctx, buf, offs, len(docs))
        assert c["stage1_status"] == 0  # (every document passes stage 1: the plain pass is accepted)
        n_bad = 0
        for k, d in enumerate(docs):
            want = O.parse(d + b"\n")
            assert int(err[k]) == want.error, (k, d, int(err[k]), want.error)
            if want.error:
                n_bad += 1
            else:
                assert O.Parsed(tape[int(to[k]):int(to[k + 1])], strings, 0, 0, 0).to_python() == want.to_python(), (k, d)
        assert n_bad >= 400 and c["failed_documents"] == n_bad
    finally:
        ctx.close()


def test_accepted_batch_with_a_large_and_a_deep_document(twitter):
    """Documents the token walker's per-document assumptions do not cover, inside an ACCEPTED batch (all pass stage 1, newline
    separators): one large document (twitter.json: 864 chunks of structurals through the ring), nesting beyond the 64 levels of
    the register stack (handed to the exact walker), nesting beyond max_depth, a document of one token -- every document against
    the oracle, tapes at their final addresses."""
    import simdjson_java_amd as S
    rng = random.Random(11)
    small = _small_docs(rng, 300)
    deep70 = b"[" * 70 + b"1" + b"]" * 70
    deep1100 = b"[" * 1100 + b"]" * 1100
    docs = small[:100] + [twitter.rstrip(b"\n")] + small[100:200] + [deep70, b"7", b'"root string"', deep1100, b"[]", b"{}"] + small[200:]
    ctx = S.Context(0, 4 << 20)
    try:
        buf = b"".join(d + b"\n" for d in docs)
        offs = np.concatenate([[0], np.cumsum([len(d) + 1 for d in docs])]).astype(np.uint64)
        c, tape, to, err, strings, io, idx = _run_shard(ctx, buf, offs, len(docs))
        assert c["stage1_status"] == 0
        for k, d in enumerate(docs):
            want = O.parse(d + b"\n")
            assert int(err[k]) == want.error, (k, d[:40], int(err[k]), want.error)
            if want.error == 0:
                assert O.Parsed(tape[int(to[k]):int(to[k + 1])], strings, 0, 0, 0).to_python() == want.to_python(), (k, d[:40])
        assert c["failed_documents"] == 1 and c["host_documents"] == 0  # (the 1,100-level document: maxDepth 1024)
    finally:
        ctx.close()


@pytest.mark.parametrize("sep", [b"\n", b" "], ids=["lf", "space"])
def test_repaired_batch_with_documents_of_every_size(twitter, sep):
    """The repair stage's sanitized copy is made by the verdict pass -- a row of 16 lanes per document, 1 KiB per trip, the copy's
    stores transposed through LDS, a document's last chunk stored byte-precisely, a failing document blanked behind its last
    trip.  Sizes that stress it: 1 .. 80 bytes (every length modulo 16 and 64), exactly 64 / 1024 / 1025 bytes, a large document
    (twitter.json: 617 trips of its row), failing documents of every size in between -- a copy with one wrong byte shows up as a
    wrong tree or a wrong verdict of the document it belongs to.  Through the call for rejected batches, every document against
    the oracle."""
    import simdjson_java_amd as S
    rng = random.Random(23)
    docs = []
    for n in list(range(1, 81)) + [127, 128, 129, 1000, 1023, 1024, 1025, 4097]:
        body = b"[" + b",".join(b"%d" % rng.randrange(10 ** rng.randint(1, 6)) for _ in range(n))
        good = (body[:max(n - 1, 1)].rstrip(b",-") if n > 2 else b"[") + b"]"
        docs.append(good)
        docs.append(b'["' + b"x" * max(n - 3, 0) + b'"]')
        if n % 3 == 0:
            docs.append(b'["' + b"y" * n)                      # unclosed string of that size: blanked
        if n % 5 == 0:
            docs.append(b'["' + b"z" * (n // 2) + b"\xff" + b'"]')  # invalid UTF-8: blanked
    docs.insert(40, twitter.rstrip(b"\n"))
    docs.insert(90, b'{"k":"' + b"w" * 5000 + b'\x01"}')         # a control character inside a long string: blanked
    ctx = S.Context(0, 4 << 20)
    try:
        buf = b"".join(d + sep for d in docs)
        offs = np.concatenate([[0], np.cumsum([len(d) + len(sep) for d in docs])]).astype(np.uint64)
        c, tape, to, err, strings, io, idx = _run_shard(ctx, buf, offs, len(docs), exact=False, want_rejected=True)
        n_bad = 0
        for k, d in enumerate(docs):
            want = O.parse(d + sep)
            assert int(err[k]) == want.error, (k, d[:40], len(d), int(err[k]), want.error)
            if want.error:
                n_bad += 1
                if 1 <= want.error <= 3:
                    assert int(to[k + 1] - to[k]) == 2, (k, len(d))  # (repaired: the tapes were laid out in advance)
            else:
                got_idx = idx[int(io[k]):int(io[k + 1])].astype(np.int64) - int(offs[k])
                assert np.array_equal(got_idx, O.stage1(d)[0].astype(np.int64)), (k, len(d))
                assert O.Parsed(tape[int(to[k]):int(to[k + 1])], strings, 0, 0, 0).to_python() == want.to_python(), (k, len(d))
        assert c["failed_documents"] == n_bad >= 40
    finally:
        ctx.close()


def test_accepted_batch_delimiters_equal_the_per_document_passes():
    """k_doc_prepare (accepted plain pass) reads a document's index range, first string and tape slot off stage 1's per-block
    side outputs instead of searching / packing.  Layouts that stress it: documents that begin exactly on a block boundary, many
    documents inside one block, dense structurals (more than 16 in front of a boundary inside its block), long documents, strings
    across boundaries.  index_offsets, doc_string_offsets and (all documents valid) tape_offsets must be those of the three
    separate calls (per-document passes + packing behind the walk)."""
    import torch
    import simdjson_java_amd as S
    from simdjson_java_amd import sharding
    rng = random.Random(77)
    docs = []
    for i in range(3000):
        r = rng.random()
        if r < 0.25:
            docs.append(rng.choice([b"1", b"[]", b"{}", b'""', b"[1]", b'"a"', b"null", b"-0", b'{"a":1}']))
        elif r < 0.40:
            n = rng.randrange(1, 80)
            docs.append(b"[" + b",".join(b"1" for _ in range(n)) + b"]")                 # a structural every byte
        elif r < 0.50:
            docs.append(b'["' + b"x" * rng.randrange(0, 200) + b'","' + b"\\\\" * rng.randrange(0, 40) + b'"]')
        elif r < 0.55:
            d = b'{"k":[' + b",".join(b'{"a":"b","c":[1,2.5,true,null]}' for _ in range(rng.randrange(50, 300))) + b"]}"
            docs.append(d)
        else:
            docs.extend(_small_docs(rng, 1))
        if r > 0.97:  # pad so that the NEXT document begins on a 64-byte boundary
            tot = sum(len(d) + 1 for d in docs)
            pad = (-tot - 3) % 64
            docs.append(b"[" + b" " * pad + b"]")
    ctx = S.Context(0, 1 << 20)
    try:
        buf, offs = _pack(docs)
        shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
        shard.step(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        c = shard.check()
        assert c["stage1_status"] == 0 and c["failed_documents"] == 0, c
        got_io = shard.index_offsets.cpu().numpy()
        got_dso = shard.doc_string_offsets.cpu().numpy()
        got_to = shard.tape_offsets.cpu().numpy()
        got_tape = shard.tape.cpu().numpy().view(np.uint64)[:int(got_to[-1])]
        # the three separate calls on the same context (isolated stage 1 is exact whatever the plain pass does)
        import tests.test_gpu_walk as TW
        tapes, strings, errors = TW.gpu_walk(ctx, docs)
        assert not errors.any()
        want_to = np.concatenate([[0], np.cumsum([t.size for t in tapes])])
        assert np.array_equal(got_to, want_to)
        assert np.array_equal(got_tape, np.concatenate(tapes))
        assert bytes(shard.sb[:c["string_bytes"]].cpu().numpy()) == strings
        # index / string offsets: recomputed from the oracle
        want_io, want_dso, si, so = [0], [], 0, 0
        for d in docs:
            idx, st = O.stage1(d + b"\n")
            assert st == 0
            want_dso.append(so)
            p = O.parse(d + b"\n")
            so += len(p.strings)
            si += idx.size
            want_io.append(si)
        want_dso.append(so)
        assert np.array_equal(got_io, np.asarray(want_io))
        assert np.array_equal(got_dso, np.asarray(want_dso))
    finally:
        ctx.close()


def test_optimistic_entry_point_equals_the_exact_one_on_an_accepted_batch():
    """sjmi_parse_batch_device_optimistic (eight queue entries) against sjmi_parse_batch_device (everything queued) on NDJSON
    that qualifies: every output array byte for byte, the three result records, nothing flagged."""
    import torch
    import simdjson_java_amd as S
    from simdjson_java_amd import sharding
    rng = random.Random(77)
    docs = _small_docs(rng, 4000) + [b"[1 1]", b'{"a":tru}', b'["\\q"]']  # (stage-2 / string errors do not reject a batch)
    rng.shuffle(docs)
    buf = b"".join(d + b"\n" for d in docs)
    offs = np.concatenate([[0], np.cumsum([len(d) + 1 for d in docs])]).astype(np.uint64)
    ctx = S.Context(0, 1 << 20)
    try:
        out = []
        for exact in (False, True):
            shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
            for t in (shard.idx, shard.sb, shard.tape):
                t.zero_()
            for _ in range(2):
                shard.step(torch.cuda.current_stream().cuda_stream, exact=exact)
                torch.cuda.synchronize()
            r = shard.result.cpu().numpy().copy()
            assert not (int(r[1]) & 0x800)
            c = shard.check()
            assert getattr(shard, "rejected_steps", 0) == 0
            out.append((r, shard.idx[:c["structurals"] + 1].cpu().numpy(), shard.index_offsets.cpu().numpy(), shard.doc_status.cpu().numpy(),
                        shard.sb[:c["string_bytes"]].cpu().numpy(), shard.doc_string_offsets.cpu().numpy(),
                        shard.tape_offsets.cpu().numpy(), shard.doc_errors.cpu().numpy(), shard.tape.cpu().numpy(), c))
        a, b = out
        assert a[9] == b[9] and a[9]["failed_documents"] == 3
        for i in range(8):
            assert np.array_equal(a[i], b[i]), i
        to, err = a[6], a[7]
        for k in range(len(docs)):  # (a failing document's slot holds unspecified words: compare the tapes of the others)
            if err[k] == 0:
                assert np.array_equal(a[8][int(to[k]):int(to[k + 1])], b[8][int(to[k]):int(to[k + 1])]), k
    finally:
        ctx.close()


_TOKEN_WALK_OFF = r"""
import random, sys
import numpy as np, torch
sys.path.insert(0, %r)
import simdjson_java_amd as S
from simdjson_java_amd import sharding
from oracle import oracle as O
from tests.test_gpu_batch import _small_docs
rng = random.Random(31)
docs = _small_docs(rng, 1500) + [b"[1 1]", b'{"a":tru}', b'["\\q"]', b"[" * 70 + b"]" * 70, b"7"]
rng.shuffle(docs)
buf = b"".join(d + b"\n" for d in docs)
offs = np.concatenate([[0], np.cumsum([len(d) + 1 for d in docs])]).astype(np.uint64)
ctx = S.Context(0, 1 << 20)
shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
for _ in range(2):
    shard.step(torch.cuda.current_stream().cuda_stream, exact=True)
    torch.cuda.synchronize()
c = shard.check()
to = shard.tape_offsets.cpu().numpy(); tape = shard.tape.cpu().numpy().view(np.uint64); err = shard.doc_errors.cpu().numpy()
strings = bytes(shard.sb[:c["string_bytes"]].cpu().numpy())
bad = 0
for k, d in enumerate(docs):
    want = O.parse(d + b"\n")
    assert int(err[k]) == want.error, (k, d[:40], int(err[k]), want.error)
    if want.error:
        bad += 1
    else:
        assert O.Parsed(tape[int(to[k]):int(to[k + 1])], strings, 0, 0, 0).to_python() == want.to_python(), k
assert c["failed_documents"] == bad == 3, (c, bad)
ctx.close()
print("TOKEN_WALK_OFF_OK", len(docs))
"""


def test_whole_batch_through_the_exact_walker_in_a_fresh_process():
    """SJMI_TOKEN_WALK=0 (read once per process) puts k_coop_walk<false> back on EVERY document of a batch -- the dispatch branch
    of walk.hip that the suite otherwise only reaches through the token walker's list.  A fresh process with the switch set: the
    exact entry point on an accepted batch with grammar errors, a string error, nesting beyond the LDS stack and a scalar root,
    every document against the oracle."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    env = dict(os.environ, SJMI_TOKEN_WALK="0")
    out = subprocess.run([sys.executable, "-c", _TOKEN_WALK_OFF % ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "TOKEN_WALK_OFF_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


_REPAIR_OFF = r"""
import random, sys
import numpy as np, torch
sys.path.insert(0, %r)
import simdjson_java_amd as S
from simdjson_java_amd import sharding
from oracle import oracle as O
from tests.test_gpu_batch import _small_docs
rng = random.Random(33)
docs = _small_docs(rng, 1500) + [b"[1 1]", b'["abc', bytes([0x5B, 0x22, 0xC3, 0x22, 0x5D]), b'["a\x01"]', b'def"]', b"7", b""]
rng.shuffle(docs)
ctx = S.Context(0, 1 << 20)
for sep in (b"\n", b" "):
    buf = b"".join(d + sep for d in docs)
    offs = np.concatenate([[0], np.cumsum([len(d) + len(sep) for d in docs])]).astype(np.uint64)
    for entry in ("optimistic", "exact"):
        shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
        for _ in range(2):
            shard.step(torch.cuda.current_stream().cuda_stream, exact=(entry == "exact"))
            torch.cuda.synchronize()
        c = shard.check()
        to = shard.tape_offsets.cpu().numpy(); tape = shard.tape.cpu().numpy().view(np.uint64); err = shard.doc_errors.cpu().numpy()
        strings = bytes(shard.sb[:c["string_bytes"]].cpu().numpy())
        bad = 0
        for k, d in enumerate(docs):
            want = O.parse(d + sep)
            assert int(err[k]) == want.error, (sep, entry, k, d[:40], int(err[k]), want.error)
            if want.error:
                bad += 1
                assert to[k + 1] == to[k]  # (the per-document passes: tapes packed behind the walk, a failing document's range is empty)
            else:
                assert O.Parsed(tape[int(to[k]):int(to[k + 1])], strings, 0, 0, 0).to_python() == want.to_python(), k
        assert c["failed_documents"] == bad == 6, (c, bad)
ctx.close()
print("REPAIR_OFF_OK", len(docs))
"""


def test_rejected_batches_without_the_repair_stage_in_a_fresh_process():
    """SJMI_BATCH_REPAIR=0 (read once per process): a rejected batch goes from the plain pass straight to the per-document
    passes, as before round 6 -- the path that is otherwise only reached by a batch the repair stage cannot take either (a scalar
    running on across a document boundary).  Documents failing stage 1 in every way and stage 2, newline and space separators,
    both entry points, every document against the oracle."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    env = dict(os.environ, SJMI_BATCH_REPAIR="0")
    out = subprocess.run([sys.executable, "-c", _REPAIR_OFF % ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "REPAIR_OFF_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_the_repair_stage_takes_what_it_can_and_only_that():
    """Which rejected batches the repair stage (sjmi_parse_batch_device_rejected, stage B of the pipeline) takes -- told apart by the
    tape slot of a document that fails stage 1: two words when the tapes were laid out in advance (repaired), empty when they were
    packed behind the per-document passes.  Separated by '\\n' or ' ' or not at all but ending in '}' / ']' / '"': repaired.  A
    scalar at the end of one document directly in front of the next one (no separator), or a document ending in a backslash:
    not.  Two unclosed strings that would cancel in a plain pass; a document of only a separator.  Everything against the oracle."""
    import simdjson_java_amd as S
    rng = random.Random(17)
    containers = [d for d in _small_docs(rng, 6000) if d[:1] in (b"{", b"[")][:1500]
    bad = [b'["abc', b'def"]', bytes([0x5B, 0x22, 0xC3, 0x22, 0x5D]), b'["a\x01b"]', b'{"k":"v']
    ctx = S.Context(0, 1 << 20)
    try:
        for name, tail, sep, repaired in (("newline", [], b"\n", True), ("space", [], b" ", True), ("none, containers only", [], b"", True),
                                          ("none, root strings", [b'"s"', b'"t"'], b"", True),
                                          ("none, a scalar in front of a document", [b"12", b"[3]"], b"", False),
                                          ("none, a trailing backslash", [b"[1]\\", b'"x"'], b"", False),
                                          ("newline, empty documents", [b"", b""], b"\n", True)):
            docs = list(containers)
            for i, b in enumerate(bad):
                docs.insert(100 + 200 * i, b)
            docs += tail
            buf = b"".join(d + sep for d in docs)
            offs = np.concatenate([[0], np.cumsum([len(d) + len(sep) for d in docs])]).astype(np.uint64)
            c, tape, to, err, strings, io, idx = _run_shard(ctx, buf, offs, len(docs), exact=False, want_rejected=True)
            n_bad = 0
            for k, d in enumerate(docs):
                want = O.parse(d + sep) if sep.strip() == b"" and sep else O.parse(d)
                assert int(err[k]) == want.error, (name, k, d[:40], int(err[k]), want.error)
                if want.error:
                    n_bad += 1
                    if 1 <= want.error <= 3:  # (failed stage 1: no structurals; the slot tells which stage took the batch)
                        assert io[k + 1] == io[k], (name, k)
                        assert int(to[k + 1] - to[k]) == (2 if repaired else 0), (name, k, int(to[k + 1] - to[k]))
                else:
                    got_idx = idx[int(io[k]):int(io[k + 1])].astype(np.int64) - int(offs[k])
                    assert np.array_equal(got_idx, O.stage1(d)[0].astype(np.int64)), (name, k)
                    assert O.Parsed(tape[int(to[k]):int(to[k + 1])], strings, 0, 0, 0).to_python() == want.to_python(), (name, k)
            assert c["failed_documents"] == n_bad >= len(bad), (name, c, n_bad)
    finally:
        ctx.close()


def test_the_call_for_rejected_batches_on_its_own():
    """sjmi_parse_batch_device_rejected as the FIRST call on a fresh context (nothing of an optimistic call's state to lean on), on a
    clean NDJSON batch and on one with failing documents, twice each (the second call reuses every workspace): outputs per document
    equal to the oracle's, and equal to the exact entry's on the same batch."""
    import torch
    import simdjson_java_amd as S
    from simdjson_java_amd import sharding
    rng = random.Random(71)
    good = _small_docs(rng, 2500)
    for bad in ([], [b'["abc', bytes([0x5B, 0x22, 0xC3, 0x22, 0x5D]), b"[1 1]", b'{"k":"v\x02"}']):
        docs = list(good)
        for i, b in enumerate(bad):
            docs.insert(300 + 500 * i, b)
        buf = b"".join(d + b"\n" for d in docs)
        offs = np.concatenate([[0], np.cumsum([len(d) + 1 for d in docs])]).astype(np.uint64)
        outs = []
        for entry in ("rejected", "exact"):
            ctx = S.Context(0, 1 << 20)
            try:
                shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
                for _ in range(2):
                    shard.step(torch.cuda.current_stream().cuda_stream, exact=(entry == "exact"), rejected=(entry == "rejected"))
                    torch.cuda.synchronize()
                st1 = int(shard.result.cpu().numpy()[1]) & 0xFFFFFFFF
                assert not (st1 & 0x800), (entry, hex(st1))
                c = shard.check()
                to = shard.tape_offsets.cpu().numpy()
                tape = shard.tape.cpu().numpy().view(np.uint64)
                err = shard.doc_errors.cpu().numpy()[:len(docs)]
                strings = bytes(shard.sb[:c["string_bytes"]].cpu().numpy())
                n_bad = 0
                for k, d in enumerate(docs):
                    want = O.parse(d + b"\n")
                    assert int(err[k]) == want.error, (entry, k, d[:40], int(err[k]), want.error)
                    if want.error:
                        n_bad += 1
                    else:
                        assert O.Parsed(tape[int(to[k]):int(to[k + 1])], strings, 0, 0, 0).to_python() == want.to_python(), (entry, k)
                assert c["failed_documents"] == n_bad == len(bad), (entry, c)
                outs.append((err.copy(), c["structurals"], c["string_bytes"], strings))
            finally:
                ctx.close()
        assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1:] == outs[1][1:]


def test_pipeline_in_safe_mode_with_a_faked_timeout_and_with_misaligned_buffers():
    """The corners of the re-ordered pipeline (round 5): (a) SAFE liveness mode -- no scanner, so the plain pass's record comes by a
    queued copy and the string workspace is still zeroed by the workers; (b) a FAST plain pass that reports a tripped spin bound
    (debug flag 16): the batch is rejected on the device (stage-1 status != 0) -- the optimistic call says SJMI_ST_REJECTED, the
    exact call decides every document by the per-document passes; (c) a misaligned batch buffer: the plain pass cannot even be
    tried -- rejected / per-document passes.  Every case against the oracle."""
    import torch
    import simdjson_java_amd as S
    from simdjson_java_amd import sharding
    rng = random.Random(41)
    docs = _small_docs(rng, 1200) + [b"[1 1]"]
    buf = b"".join(d + b"\n" for d in docs)
    offs = np.concatenate([[0], np.cumsum([len(d) + 1 for d in docs])]).astype(np.uint64)
    want = [O.parse(d + b"\n") for d in docs]

    def verify(shard, label):
        c = shard.check()
        to = shard.tape_offsets.cpu().numpy()
        tape = shard.tape.cpu().numpy().view(np.uint64)
        err = shard.doc_errors.cpu().numpy()
        strings = bytes(shard.sb[:c["string_bytes"]].cpu().numpy())
        for k, w in enumerate(want):
            assert int(err[k]) == w.error, (label, k)
            if not w.error:
                assert O.Parsed(tape[int(to[k]):int(to[k + 1])], strings, 0, 0, 0).to_python() == w.to_python(), (label, k)
        assert c["failed_documents"] == 1, (label, c)
        return c

    st = torch.cuda.current_stream().cuda_stream
    ctx = S.Context(0, 1 << 20)
    try:
        # (a) SAFE mode
        ctx.set_tile_mode(True)
        for exact in (False, True):
            shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
            for _ in range(2):
                shard.step(st, exact=exact)
                torch.cuda.synchronize()
            assert not (int(shard.result.cpu().numpy()[1]) & 0x800)
            verify(shard, "safe exact=%s" % exact)
        ctx.set_tile_mode(False)
        # (b) a faked liveness trip of the FAST plain pass
        ctx.debug_set_flags(16)
        shard = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
        shard.step(st)
        torch.cuda.synchronize()
        assert int(shard.result.cpu().numpy()[1]) & 0x800, "the optimistic call must report the rejected plain pass"
        ctx.debug_set_flags(0)
        shard2 = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
        ctx.debug_set_flags(16)
        shard2.step(st, exact=True)
        torch.cuda.synchronize()
        ctx.debug_set_flags(0)
        verify(shard2, "faked timeout, exact call")
        # (c) a batch buffer that is not 16-byte aligned
        shard3 = sharding.BatchShard(ctx, buf, offs, torch.device("cuda", 0))
        moved = torch.zeros(shard3.n + 256, dtype=torch.uint8, device="cuda")
        moved[1:1 + shard3.n] = shard3.buf[:shard3.n]
        shard3.buf = moved[1:]
        assert shard3.buf.data_ptr() % 16 != 0
        shard3.step(st)
        torch.cuda.synchronize()
        assert int(shard3.result.cpu().numpy()[1]) & 0x800
        verify(shard3, "misaligned (check() makes the exact call)")
        assert shard3.rejected_steps == 1
    finally:
        ctx.close()
