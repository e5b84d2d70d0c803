"""The on-demand front end's skip table (sjmi_match_brackets, csrc/coop_walk.hip k_coop_match) against the reference's own
way of leaving a value: OnDemandJsonIterator.skipChild(parentDepth) (OnDemandJsonIterator.java:43-81) restated here as the
scan it is -- from EVERY read position of the reference files and for every number of containers to leave, the table
lookup must land on the structural the scan lands on (or both must run out of closing brackets)."""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import load_fixture

pytestmark = pytest.mark.gpu

NONE, UNKNOWN = 0xFFFFFFFF, 0xFFFFFFFE


def skip_child_scan(doc, idx, r, depth, parent_depth):
    """OnDemandJsonIterator.skipChild(int parentDepth), :47-81, over the structural positions idx (BitIndexes cursor r).
    -> (r, depth) after the call, or None for 'Not enough close braces.'"""
    n = len(idx)

    def byte(k):  # BitIndexes reads its 0 sentinel past the end (BitIndexes.java:82-96)
        return doc[idx[k]] if k < n else doc[0]
    if depth <= parent_depth:
        return r, depth
    if r >= n:
        return None
    ch = byte(r)
    r += 1
    if ch in b"[{:,":
        pass
    elif ch == 0x22 and byte(r) == 0x3A:
        r += 1
    else:
        depth -= 1
        if depth <= parent_depth:
            return r, depth
    while r < n:
        ch = byte(r)
        r += 1
        delta = 1 if ch in b"[{" else (-1 if ch in b"]}" else 0)
        depth += delta
        if delta < 0 and depth <= parent_depth:
            return r, depth
    return None


def skip_child_table(doc, idx, up, match, r, depth, parent_depth):
    """The same call with the skip table: the first structural is looked at exactly as the reference does, then k containers
    are left by k - 1 climbs through up[] and one jump through match[]."""
    n = len(idx)
    if depth <= parent_depth:
        return r, depth
    if r >= n:
        return None
    ch = doc[idx[r]]
    q = r + 1
    if ch in b"[{:,":
        pass
    elif ch == 0x22 and (doc[idx[q]] if q < n else doc[0]) == 0x3A:
        q += 1
    else:
        depth -= 1
        if depth <= parent_depth:
            return q, depth
    # the scan now counts brackets from q on: it stops at the closing bracket that takes the depth to parent_depth
    k = depth - parent_depth
    if q >= n:
        return None
    # the container position q lies in: the bracket just consumed if the first structural was an opening one (the
    # reference breaks out of its switch without counting it, so its closing bracket is the first -1 the scan meets)
    e = int(up[q]) if ch not in b"[{" else r
    if e in (NONE, UNKNOWN):
        return "unknown" if e == UNKNOWN else None
    for _ in range(k - 1):
        e = int(up[e])
        if e in (NONE, UNKNOWN):
            return "unknown" if e == UNKNOWN else None
    m = int(match[e])
    if m in (NONE, UNKNOWN):
        return "unknown" if m == UNKNOWN else None
    return m + 1, parent_depth


def _tables(ctx, doc):
    idx, st = ctx.stage1(doc)
    up, match = ctx.match_brackets(idx.size)
    return [int(x) for x in idx], up, match


@pytest.fixture(scope="module")
def ctx():
    import simdjson_java_amd as S
    c = S.Context(device=0, capacity=8 * 1024 * 1024)
    yield c
    c.close()


def _check_table_against_stack(doc, idx, up, match):
    """independent of skipChild: up / match against a plain bracket stack"""
    stack = []
    for i, p in enumerate(idx):
        ch = doc[p]
        if ch in b"]}":
            assert stack, "test documents are balanced"
            o = stack[-1]
            assert int(up[i]) == o and int(match[i]) == o and int(match[o]) == i, i
            stack.pop()
        else:
            want = stack[-1] if stack else NONE
            assert int(up[i]) == want, (i, int(up[i]), want)
            if ch in b"[{":
                stack.append(i)
            else:
                assert int(match[i]) == want
    for o in stack:
        assert int(match[o]) == NONE


@pytest.mark.parametrize("name", ["twitter.json", "github_events.json", "wide_bench.json"])
def test_skip_child_from_every_position(ctx, name):
    doc = load_fixture(name)
    idx, up, match = _tables(ctx, doc)
    _check_table_against_stack(doc, idx, up, match)
    rng = random.Random(7)
    positions = range(len(idx)) if len(idx) < 1500 else sorted(rng.sample(range(len(idx)), 1500))
    checked = 0
    for r in positions:
        for k in (1, 2, 3, 5):
            depth = 10  # the iterator's own bookkeeping: only depth - parentDepth matters to the scan
            want = skip_child_scan(doc, idx, r, depth, depth - k)
            got = skip_child_table(doc, idx, up, match, r, depth, depth - k)
            assert got != "unknown"
            assert got == want, (r, k, got, want, bytes(doc[idx[r]:idx[r] + 12]))
            checked += 1
    assert checked >= 6000


def test_nesting_steps_and_unbalanced_documents(ctx):
    rng = random.Random(8)

    def nested(d):
        if d == 0:
            return rng.choice(["1", '"x"', "true", "[]", "{}"])
        if rng.random() < 0.5:
            return "[" + ",".join(nested(d - 1) for _ in range(rng.randint(1, 4))) + "]"
        return "{" + ",".join('"k%d":%s' % (i, nested(d - 1)) for i in range(rng.randint(1, 4))) + "}"
    for doc in [nested(7).encode() for _ in range(6)] + [("[" * d + "1" + "]" * d).encode() for d in (1, 63, 64)] + \
               [b"[" + b"[1,[2,[3]]]," * 300 + b"0]"]:
        idx, up, match = _tables(ctx, doc)
        _check_table_against_stack(doc, idx, up, match)
        for r in range(0, len(idx), 7):
            for k in (1, 2, 4):
                assert skip_child_table(doc, idx, up, match, r, 9, 9 - k) == skip_child_scan(doc, idx, r, 9, 9 - k), (r, k)
    # never-closed brackets: NONE, and the scan runs out of closing brackets exactly where the table says so
    doc = b'{"a":[1,2,{"b":[3,4'
    idx, up, match = _tables(ctx, doc)
    assert [int(match[i]) for i, p in enumerate(idx) if doc[p] in b"[{"] == [NONE] * 4
    for r in range(len(idx)):
        for k in (1, 2):
            assert skip_child_table(doc, idx, up, match, r, 5, 5 - k) == skip_child_scan(doc, idx, r, 5, 5 - k)
    # a closing bracket without an opening one: everything from there on is left to the scan
    doc = b"[1,2]] [3,[4]]"
    idx, up, match = _tables(ctx, doc)
    bad = [i for i, p in enumerate(idx) if p == 5][0]
    assert all(int(x) == UNKNOWN for x in up[bad:]) and all(int(x) != UNKNOWN for x in up[:bad])
    # beyond the 64 levels of the per-wave stack: marked, not guessed
    doc = ("[" * 70 + "1" + "]" * 70).encode()
    idx, up, match = _tables(ctx, doc)
    assert int(up[69]) == UNKNOWN and int(up[30]) == 29 and int(match[0]) == len(idx) - 1


# ---- the on-demand cursor itself (csrc/host/ondemand.h over the C ABI: sjmi_parser_ondemand_init + sjmi_od_*) ----
from oracle import ondemand as OD  # noqa: E402
from tests.golden.ondemand_vectors import VECTORS  # noqa: E402
from tests.ondemand_common import OracleIterator, fuzz_walk, run_oracle, walk_document  # noqa: E402


@pytest.fixture(scope="module")
def parser():
    import simdjson_java_amd as S
    p = S.SimdJsonParser(capacity=8 * 1024 * 1024)
    yield p
    p.close()


def _run_parser(parser, doc, length, schema, table):
    import simdjson_java_amd as S
    try:
        return "ok", walk_document(parser.ondemand(doc, length, skip_table=table), schema)
    except S.JsonParsingException as e:
        return "error", str(e)


@pytest.mark.parametrize("table", [False, True], ids=["scan", "skip-table"])
def test_reference_schema_vectors_through_the_c_abi(parser, table):
    """The 287 inputs of tests/golden/ondemand_vectors.py (values / messages asserted by the reference's own
    *SchemaBasedParsingTest classes) through GPU stage 1 (+ the GPU skip table) and the C ABI cursor."""
    for (j, length, schema, value, message) in VECTORS:
        doc = j.encode("utf-8")
        n = len(doc) if length is None else length
        kind, got = _run_parser(parser, doc, n, schema, table)
        if message is not None:
            assert (kind, got) == ("error", message), (j, schema, got)
        else:
            assert kind == "ok" and got == value, (j, schema, got)


@pytest.mark.parametrize("table", [False, True], ids=["scan", "skip-table"])
def test_parse_and_select_twitter_on_demand(parser, table):
    """BenchmarkCorrectnessTest.java:23-55 (schemaBasedSimdJsonParser): the screen names of the users with default_profile,
    selected on demand -- 86 of them, the same set the full parse + JsonValue walk finds (tests/test_gpu_parse.py)."""
    import json
    doc = load_fixture("twitter.json")
    schema = ("object", {"statuses": ("array", ("object", {"user": ("object", {"default_profile": "boolean", "screen_name": "String"})}))})
    kind, got = _run_parser(parser, doc, len(doc), schema, table)
    assert kind == "ok"
    names = {u["user"]["screen_name"] for u in got["statuses"] if u["user"]["default_profile"]}
    assert len(names) == 86
    assert names == {s["user"]["screen_name"].encode() for s in json.loads(doc)["statuses"] if s["user"]["default_profile"]}
    idx, _ = O.stage1(doc)
    assert run_oracle(doc, len(doc), idx, schema) == (kind, got)


@pytest.mark.parametrize("table", [False, True], ids=["scan", "skip-table"])
def test_on_demand_fuzz_traces(parser, table):
    """Random (often broken) documents, schema-less walk with seeded skips / wrong-typed reads / early exits: the trace of the
    C ABI cursor (GPU indexes, GPU skip table) equals the trace of the restated reference, exception message included."""
    import simdjson_java_amd as S
    from tests.test_host_ondemand import _random_doc
    rng = random.Random(77)
    walked = 0
    for _ in range(1500):
        doc = _random_doc(rng).encode("utf-8")
        idx, st = O.stage1(doc)
        seed = rng.getrandbits(32)
        want = []
        try:
            if st:
                raise OD.JsonParsingException(O.error_message(1 if st & 1 else (2 if st & 2 else 3)))
            fuzz_walk(OracleIterator(doc, len(doc), idx), random.Random(seed), want)
            want.append("done")
        except OD.JsonParsingException as e:
            want.append(("raised", str(e)))
        got = []
        try:
            fuzz_walk(parser.ondemand(doc, skip_table=table), random.Random(seed), got)
            got.append("done")
        except S.JsonParsingException as e:
            got.append(("raised", str(e)))
        assert got == want, (doc, seed, got[-3:], want[-3:])
        walked += 1
    assert walked == 1500


def test_skip_table_chunk_boundaries(ctx):
    """The chunk-parallel skip table (k_match_summary -> scan kernels -> k_coop_match<true>) against a plain bracket stack
    at structural counts around its switches (1,024; multiples of 128 and 512; 262,144), with nested brackets across the
    chunk boundaries, and for documents that never close / close too often (SJMI_MATCH_NONE / _UNKNOWN)."""
    rng = random.Random(12)

    def nested(n):
        out, depth = [], 0
        for i in range(n):
            r = rng.random()
            if r < 0.25 and depth < 40:
                out.append(rng.choice("[{") if False else "[")
                depth += 1
            elif r < 0.45 and depth > 0:
                out.append("]")
                depth -= 1
            else:
                out.append("0")
        return out, depth
    for n in (600, 1500, 5000, 70000, 300000):
        toks, depth = nested(n)
        body = []
        for i, t in enumerate(toks):
            body.append(t)
            nxt = toks[i + 1] if i + 1 < len(toks) else "]"
            if t != "[" and nxt != "]":
                body.append(",")
        doc = ("[" + "".join(body) + "]" * (depth + 1)).encode()
        idx, up, match = _tables(ctx, doc)
        _check_table_against_stack(doc, idx, up, match)
    for s in (1023, 1025, 1153, 8193, 262145, 262144 + 513):
        doc = ("[" + "[0]," * ((s - 3) // 4) + "0]").encode()
        idx, up, match = _tables(ctx, doc)
        _check_table_against_stack(doc, idx, up, match)
        # never closed: the root bracket (and the last "[0" cut open) keep SJMI_MATCH_NONE
        cut = doc[:-3]
        idx, up, match = _tables(ctx, cut)
        assert int(match[0]) == NONE
        # closed too often, far from the start: everything from the stray bracket on is SJMI_MATCH_UNKNOWN
        stray = doc[:-1] + b"]],[1,2]"   # ... 0 ] ] , [ 1 , 2 ]  : the second "]" has no opening bracket
        idx, up, match = _tables(ctx, stray)
        k = len(idx) - 7                  # the stray bracket
        assert int(match[0]) == k - 1 and int(up[k - 1]) == 0
        assert all(int(up[i]) == UNKNOWN and int(match[i]) == UNKNOWN for i in range(k, len(idx)))


def test_reference_float_vectors_through_the_c_abi(parser):
    """tests/golden/float_vectors.json (the reference's binary32 tests) through GPU stage 1 and sjmi_od_get_float."""
    import json
    import os
    vs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "float_vectors.json")))
    for v in vs:
        doc = v["input"].encode()
        want = OD.float_bits(OD.float32_of(v["input"]))
        if v["bits"] is not None:
            assert want == int(v["bits"], 16)
        for schema in ("float", "Float"):
            kind, got = _run_parser(parser, doc, len(doc), schema, False)
            assert kind == "ok" and OD.float_bits(got) == want, (v, got)
    assert _run_parser(parser, b"null", 4, "Float", False) == ("ok", None)
    assert _run_parser(parser, b"null", 4, "float", False) == ("error", "Invalid number. Minus has to be followed by a digit.")
    assert _run_parser(parser, b"12", 2, "float", False) == ("error", "Invalid floating-point number. Fraction or exponent part is missing.")
