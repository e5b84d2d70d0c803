"""The on-demand front end through the GPU and the C ABI (sjmi_parser_ondemand_init + sjmi_od_*): the C++ mirror of
OnDemandJsonIterator (csrc/host/ondemand.h) over GPU stage 1 against the Python restatement of the reference's cursor
(oracle/ondemand.py, pinned by the 287 vectors the reference's *SchemaBasedParsingTest classes assert).  skip_child_scan is
OnDemandJsonIterator.skipChild(parentDepth) (OnDemandJsonIterator.java:43-81) restated as the scan it is.
(Rounds 2-4 also tested a GPU skip table here -- removed in round 5: it never paid)"""
import random

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import load_fixture

pytestmark = pytest.mark.gpu



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


def _run_parser(parser, doc, length, schema):
    import simdjson_java_amd as S
    try:
        return "ok", walk_document(parser.ondemand(doc, length), schema)
    except S.JsonParsingException as e:
        return "error", str(e)


def test_reference_schema_vectors_through_the_c_abi(parser):
    """The 287 inputs of tests/golden/ondemand_vectors.py (values / messages asserted by the reference's own
    *SchemaBasedParsingTest classes) through GPU stage 1 and the C ABI cursor."""
    for (j, length, schema, value, message) in VECTORS:
        doc = j.encode("utf-8")
        n = len(doc) if length is None else length
        kind, got = _run_parser(parser, doc, n, schema)
        if message is not None:
            assert (kind, got) == ("error", message), (j, schema, got)
        else:
            assert kind == "ok" and got == value, (j, schema, got)


def test_parse_and_select_twitter_on_demand(parser):
    """BenchmarkCorrectnessTest.java:23-55 (schemaBasedSimdJsonParser): the screen names of the users with default_profile,
    selected on demand -- 86 of them, the same set the full parse + JsonValue walk finds (tests/test_gpu_parse.py)."""
    import json
    doc = load_fixture("twitter.json")
    schema = ("object", {"statuses": ("array", ("object", {"user": ("object", {"default_profile": "boolean", "screen_name": "String"})}))})
    kind, got = _run_parser(parser, doc, len(doc), schema)
    assert kind == "ok"
    names = {u["user"]["screen_name"] for u in got["statuses"] if u["user"]["default_profile"]}
    assert len(names) == 86
    assert names == {s["user"]["screen_name"].encode() for s in json.loads(doc)["statuses"] if s["user"]["default_profile"]}
    idx, _ = O.stage1(doc)
    assert run_oracle(doc, len(doc), idx, schema) == (kind, got)


def test_on_demand_fuzz_traces(parser):
    """Random (often broken) documents, schema-less walk with seeded skips / wrong-typed reads / early exits: the trace of the
    C ABI cursor (GPU indexes) equals the trace of the restated reference, exception message included."""
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
            fuzz_walk(parser.ondemand(doc), random.Random(seed), got)
            got.append("done")
        except S.JsonParsingException as e:
            got.append(("raised", str(e)))
        assert got == want, (doc, seed, got[-3:], want[-3:])
        walked += 1
    assert walked == 1500
