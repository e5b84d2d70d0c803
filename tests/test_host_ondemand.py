"""The on-demand front end (csrc/host/ondemand.h: C++ mirror of OnDemandJsonIterator) without a
GPU: the engine ABI is stubbed (tests/host_sim/walk_sim.cpp), the structural indexes come from the oracle's stage 1,
and a schema driver / a fuzz driver make the same calls on the product's cursor
and on the Python restatement of the reference (oracle/ondemand.py) -- values, depth bookkeeping and exception messages
must agree call for call.  The restatement itself is pinned by tests/golden/ondemand_vectors.py: 287 inputs with the
values / messages the reference's own *SchemaBasedParsingTest classes assert."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import ondemand as OD
from oracle import oracle as O
from tests.conftest import load_fixture
from tests.golden.ondemand_vectors import VECTORS
from tests.ondemand_common import OracleIterator, fuzz_walk, run_oracle, walk_document

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_DIR = os.path.join(ROOT, "tests", "host_sim")
class SimException(Exception):
    pass


class SimIterator:
    """tests/host_sim/walk_sim.cpp sim_od_*: the product's OnDemandJsonIterator over indexes the test supplies"""

    def __init__(self, lib, doc, length, idx):
        self.lib = lib
        self.padded = np.frombuffer(bytes(doc[:length]) + b"\0" * 64, dtype=np.uint8)
        self.ix = np.concatenate([np.asarray(idx, dtype=np.uint32), [0]]).astype(np.uint32)
        code = C.c_int(0)
        self.h = lib.sim_od_create(self.padded.ctypes.data, length, self.ix.ctypes.data, len(idx), C.byref(code))
        if code.value:
            msg = lib.sim_od_message(self.h).decode("utf-8")
            lib.sim_od_destroy(self.h)
            self.h = None
            raise SimException(msg)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.sim_od_destroy(self.h)

    def _call(self, op, a=0, b=0):
        out, dout, ptr, n = C.c_int64(0), C.c_double(0), C.c_void_p(), C.c_uint64(0)
        rc = self.lib.sim_od_call(self.h, op, a, b, C.byref(out), C.byref(dout), C.byref(ptr), C.byref(n))
        if rc:
            raise SimException(self.lib.sim_od_message(self.h).decode("utf-8"))
        return out.value, dout.value, (C.string_at(ptr.value, n.value) if n.value and ptr.value else b""), n.value

    def depth_value(self):
        return self.lib.sim_od_depth(self.h)

    def peek_byte(self):
        return self.lib.sim_od_peek(self.h)

    def read_idx(self):
        return self.lib.sim_od_read_idx(self.h)

    def skip_child(self, parent_depth=None):
        self._call(0, -1 if parent_depth is None else parent_depth)

    def get_boolean(self, root=False, nullable=True):
        v = self._call(1, int(root), int(nullable))[0]
        return None if v == -1 else bool(v)

    def get_long(self, root=False, nullable=True, bits=64):
        v, _, _, isnull = self._call(2, int(root) | (bits << 8 if bits != 64 else 0), int(nullable))
        return None if isnull else v

    def get_double(self, root=False, nullable=True):
        _, d, _, isnull = self._call(3, int(root), int(nullable))
        return None if isnull else d

    def get_float(self, root=False, nullable=True):
        _, d, _, isnull = self._call(12, int(root), int(nullable))
        return None if isnull else d

    def get_char(self, root=False, nullable=True):
        v = self._call(13, int(root), int(nullable))[0]
        return None if v == -1 else v

    def get_string(self, root=False):
        v, _, s, _ = self._call(4, int(root))
        return None if v == -1 else s

    def get_field_name(self):
        return self._call(5)[2]

    def start_iterating_array(self, root=False):
        return self._call(6, int(root))[0]

    def next_array_element(self):
        return bool(self._call(7)[0])

    def start_iterating_object(self, root=False):
        return self._call(8, int(root))[0]

    def next_object_field(self):
        return bool(self._call(9)[0])

    def move_to_field_value(self):
        self._call(10)

    def assert_no_more_json_values(self):
        self._call(11)


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(SIM_DIR, "libwalksim.so")
    deps = [os.path.join(SIM_DIR, "walk_sim.cpp")] + [os.path.join(ROOT, "simdjson-java_amd", "csrc", "host", f)
                                                       for f in ("simdjson_parser.cpp", "simdjson_parser.h", "ondemand.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so, deps[0]])
    L = C.CDLL(so)
    L.sim_od_create.restype = C.c_void_p
    L.sim_od_create.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    L.sim_od_destroy.argtypes = [C.c_void_p]
    L.sim_od_message.restype = C.c_char_p
    L.sim_od_message.argtypes = [C.c_void_p]
    L.sim_od_call.restype = C.c_int
    L.sim_od_call.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sim_od_depth.argtypes = [C.c_void_p]
    L.sim_od_peek.argtypes = [C.c_void_p]
    L.sim_od_read_idx.restype = C.c_uint64
    L.sim_od_read_idx.argtypes = [C.c_void_p]
    L.sim_od_set.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    return L


def run_product(lib, doc, length, idx, schema):
    try:
        return "ok", walk_document(SimIterator(lib, doc, length, idx), schema)
    except SimException as e:
        return "error", str(e)


def test_oracle_restatement_is_pinned_by_the_reference_vectors():
    """oracle/ondemand.py against what the reference's own schema-based tests assert"""
    for (j, length, schema, value, message) in VECTORS:
        doc = j.encode("utf-8")
        n = len(doc) if length is None else length
        idx, _ = O.stage1(doc[:n])
        kind, got = run_oracle(doc, n, idx, schema)
        if message is not None:
            assert (kind, got) == ("error", message), (j, schema)
        else:
            assert kind == "ok" and got == value, (j, schema, got)


def test_reference_vectors_on_the_product_iterator(lib):
    for (j, length, schema, value, message) in VECTORS:
        doc = j.encode("utf-8")
        n = len(doc) if length is None else length
        idx, _ = O.stage1(doc[:n])
        kind, got = run_product(lib, doc, n, idx, schema)
        if message is not None:
            assert (kind, got) == ("error", message), (j, schema, got)
        else:
            assert kind == "ok" and got == value, (j, schema, got)


def _twitter_schema():
    # BenchmarkCorrectnessTest.java:23-55 / SimdJsonTwitter: statuses[].user{default_profile, screen_name}
    return ("object", {"statuses": ("array", ("object", {"user": ("object", {"default_profile": "boolean", "screen_name": "String"})}))})


def test_parse_and_select_twitter(lib):
    """The reference's schema-based benchmark selection (BenchmarkCorrectnessTest.schemaBasedSimdJsonParser): the users with
    default_profile == true -- the same 86 names the full parse finds."""
    doc = load_fixture("twitter.json")
    idx, st = O.stage1(doc)
    assert st == 0
    kind, got = run_product(lib, doc, len(doc), idx, _twitter_schema())
    assert kind == "ok"
    names = {u["user"]["screen_name"] for u in got["statuses"] if u["user"]["default_profile"]}
    assert len(got["statuses"]) == 100 and len(names) == 86
    assert run_oracle(doc, len(doc), idx, _twitter_schema()) == (kind, got)
    import json
    tree = json.loads(doc)  # (independent of everything here)
    assert names == {s["user"]["screen_name"].encode() for s in tree["statuses"] if s["user"]["default_profile"]}


def _random_doc(rng, depth=0):
    r = rng.random()
    if depth > 4 or r < 0.35:
        return rng.choice(['"s"', '"a\\nb"', '"é€"', '"\\u00e9\\uD83D\\uDE00"', "1", "-25", "2.5e3", "-0.125", "true", "false", "null", '""', "12345678901234567890",
                           "1e400", "tru", "01", "1.", "nul", '"\\q"', '"\\uD800x"', "[]", "{}", "0", "-", "1e", '"k":1', ",", "]", "}"])
    if r < 0.68:
        return "[" + rng.choice([",", ", ", " , ", " "]).join(_random_doc(rng, depth + 1) for _ in range(rng.randint(0, 5))) + rng.choice(["]", "]", "]", "]", "}", ""])
    return "{" + rng.choice([",", ",", ", ", " "]).join('%s%s%s' % (rng.choice(['"k%d"' % i, '"a\\tb"', '"k"', "1", ""]), rng.choice([":", ":", " : ", "", ","]),
                                                                     _random_doc(rng, depth + 1)) for i in range(rng.randint(0, 5))) + rng.choice(["}", "}", "}", "}", "]", ""])


def test_fuzz_traces_equal_the_restatement(lib):
    """Random (often broken) documents walked by the schema-less driver with seeded skips, wrong-typed reads and early
    exits: the product's trace -- every value, every depth, the message of the first exception -- equals the restatement's."""
    rng = random.Random(20260925)
    walked = errors = 0
    for it in range(3000):
        doc = _random_doc(rng).encode("utf-8")
        idx, st = O.stage1(doc)
        if st:
            continue
        seed = rng.getrandbits(32)
        traces = []
        for make in (lambda: OracleIterator(doc, len(doc), idx), lambda: SimIterator(lib, doc, len(doc), idx)):
            tr = []
            try:
                fuzz_walk(make(), random.Random(seed), tr)
                tr.append("done")
            except (OD.JsonParsingException, SimException) as e:
                tr.append(("raised", str(e)))
            traces.append(tr)
        assert traces[0] == traces[1], (doc, seed, traces[0][-3:], traces[1][-3:])
        walked += 1
        errors += traces[0][-1] != "done"
    assert walked > 2500 and 500 < errors < walked - 300


def test_skip_child_from_every_position_equals_the_reference_scan(lib):
    """From sampled structurals of the reference files: skipChild(parentDepth) leaving 1, 2, 3 containers on the product's cursor
    = OnDemandJsonIterator.skipChild restated as the scan it is (:47-81): read position and depth, including the positions from
    which the document runs out of brackets."""
    from tests.test_gpu_ondemand import skip_child_scan
    rng = random.Random(5)
    for name in ("twitter.json", "github_events.json"):
        doc = load_fixture(name)
        idx, _ = O.stage1(doc)
        ix = [int(x) for x in idx]
        for r in sorted(rng.sample(range(len(idx)), 400)):
            for k in (1, 2, 3):
                it = SimIterator(lib, doc, len(doc), idx)
                it.lib.sim_od_set(it.h, r, 10)  # (only depth - parentDepth matters to skipChild)
                want = skip_child_scan(doc, ix, r, 10, 10 - k)
                try:
                    it.skip_child(10 - k)
                    got = (it.read_idx(), it.depth_value())
                except SimException as e:
                    got = None
                    assert "Not enough close braces" in str(e)
                assert got == want, (name, r, k, got, want)


def test_reference_float_vectors(lib):
    """Every literal of the reference's binary32 tests (FloatingPointNumberSchemaBasedParsingTest: zeros, infinities, min / max
    normal and subnormal, rounding overflow, ties to even, round up / down -- tests/golden/float_vectors.json, extracted by
    make_float_vectors.py): the asserted constant where the test names one, else Float.parseFloat = the correctly rounded
    binary32, which oracle/ondemand.py computes exactly -- on the restatement and on the product's cursor."""
    import json
    vs = json.load(open(os.path.join(ROOT, "tests", "golden", "float_vectors.json")))
    assert len(vs) >= 120
    for v in vs:
        doc = v["input"].encode()
        idx, _ = O.stage1(doc)
        want = OD.float_bits(OD.float32_of(v["input"]))
        if v["bits"] is not None:
            assert want == int(v["bits"], 16), v
        for schema in ("float", "Float"):
            assert run_oracle(doc, len(doc), idx, schema) == ("ok", OD.float32_of(v["input"]))
            kind, got = run_product(lib, doc, len(doc), idx, schema)
            assert kind == "ok" and OD.float_bits(got) == want, (v, got)


def test_float_and_double_getters_on_random_literals(lib):
    """getFloat / getDouble of the product's cursor on random literals -- plain ones, exact midpoints of two adjacent binary32
    / binary64 values and their neighbours 10^-30 away, subnormal and overflow boundaries -- against exact roundings
    (binary32: rational arithmetic, oracle/ondemand.py float32_of; binary64: Python's float)."""
    import struct
    from decimal import Decimal, getcontext
    from tests.walk_common import random_number_literal
    getcontext().prec = 400
    rng = random.Random(31337)
    lits = []
    for _ in range(1500):
        lit = random_number_literal(rng)
        if not any(c in lit for c in ".eE"):
            lit += ".0"
        lits.append(lit)
    for _ in range(600):  # binary32 midpoints (and just beside them)
        bits = rng.getrandbits(23) | (rng.choice([0, 1, 2, 100, 127, 150, 253, 254]) << 23)
        lo = Decimal(struct.unpack("<f", struct.pack("<I", bits))[0])
        hi = Decimal(struct.unpack("<f", struct.pack("<I", bits + 1))[0]) if bits + 1 < 0x7F800000 else Decimal(2) ** 128
        mid = (lo + hi) / 2
        for nudge in (Decimal(0), Decimal(10) ** (mid.adjusted() - 30), -Decimal(10) ** (mid.adjusted() - 30)):
            text = format(mid + nudge, "f")
            lits.append(text if "." in text else text + ".0")
    checked_f = checked_d = 0
    for lit in lits:
        doc = lit.encode()
        idx, st = O.stage1(doc)
        assert st == 0
        kind, got = run_product(lib, doc, len(doc), idx, "float")
        assert kind == "ok" and OD.float_bits(got) == OD.float_bits(OD.float32_of(lit)), (lit, got)
        checked_f += 1
        kind, got = run_product(lib, doc, len(doc), idx, "double")
        assert kind == "ok" and OD.double_bits(got) == OD.double_bits(float(lit)), (lit, got)
        checked_d += 1
    assert checked_f == checked_d == len(lits) > 3000
