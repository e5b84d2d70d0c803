"""A C (not Python) host of the C ABI: tests/c_caller/sjmi_c_caller.c dlopens libsjmi.so like a JVM's FFM linker would and
(a) replays the call sequence of INTEGRATION.md sections 2-3 and 6 -- its counts and FNV-1a hashes of the index array, the
string buffer and the tape must be the oracle's; (b) calls sjmi_stage1 / sjmi_stage1_unescape / sjmi_parser_parse /
sjmi_stream_push on documents whose byte `len` is the first byte of a PROT_NONE page (len in {0, 1, 63, 64, 65, 4095, 4096,
631515}): the claim of include/sjmi.h that bytes >= len are never read from the caller's buffer, which is what lets the Java
binding pass an unpadded byte[]."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.conftest import ROOT, load_fixture

pytestmark = pytest.mark.gpu

SRC = os.path.join(ROOT, "tests", "c_caller", "sjmi_c_caller.c")
EXE = os.path.join(ROOT, "tests", "c_caller", "sjmi_c_caller")


def build_c_caller():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", EXE, SRC, "-ldl"])
    return EXE


def _fnv(b):
    h = 0xCBF29CE484222325
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    for x in a.tolist():
        h = ((h ^ x) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _run(mode, path):
    import simdjson_java_amd as S
    S.build()
    lib = os.path.join(ROOT, "simdjson-java_amd", "libsjmi.so")
    out = subprocess.run([build_c_caller(), lib, mode, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    return out.stdout


@pytest.mark.parametrize("name", ["twitter.json", "github_events.json"])
def test_ffm_call_sequence_from_c_equals_the_oracle(name, tmp_path):
    try:
        doc = load_fixture(name)
    except FileNotFoundError:
        pytest.skip("fixture %s not in tests/golden/data" % name)
    p = tmp_path / name
    p.write_bytes(doc)
    out = _run("replay", str(p))
    s1 = dict(re.findall(r"(\w+)=(\w+)", out.splitlines()[0]))
    s2 = dict(re.findall(r"(\w+)=(-?\w+)", out.splitlines()[1]))
    idx, st = O.stage1(doc)
    want = O.parse(doc)
    assert int(s1["count"]) == idx.size and int(s1["status"]) == st == 0
    assert int(s1["idxhash"], 16) == _fnv(idx.astype(np.uint32).tobytes())
    sb = bytes(want.strings)
    assert int(s1["string_bytes"]) == int(s1["walked_bytes"]) == len(sb) and int(s1["bad"]) == 0
    assert int(s1["sbhash"], 16) == _fnv(sb)
    assert int(s1["strings"]) == int((np.frombuffer(doc, dtype=np.uint8)[idx] == 0x22).sum())
    assert int(s2["error"]) == want.error == 0 and int(s2["tape_len"]) == len(want.tape)
    assert int(s2["tapehash"], 16) == _fnv(np.asarray(want.tape, dtype=np.uint64).tobytes())


@pytest.mark.parametrize("name", ["twitter.json", "github_events.json"])
def test_off_heap_engine_sequence_from_c_equals_the_oracle(name, tmp_path):
    """java/org/simdjson/Sjmi.java's Engine (ordinary downcalls on registered off-heap segments, input segment = staging):
    both passes -- fresh and cached device views -- give the oracle's indexes and string records"""
    doc = load_fixture(name)
    p = tmp_path / name
    p.write_bytes(doc)
    out = _run("engine", str(p))
    lines = out.splitlines()
    assert len(lines) == 2 and lines[0] == lines[1]
    s1 = dict(re.findall(r"(\w+)=(\w+)", lines[0]))
    idx, st = O.stage1(doc)
    sb = bytes(O.parse(doc).strings)
    assert int(s1["count"]) == idx.size and int(s1["status"]) == st == 0
    assert int(s1["idxhash"], 16) == _fnv(idx.astype(np.uint32).tobytes())
    assert int(s1["string_bytes"]) == int(s1["walked_bytes"]) == len(sb) and int(s1["bad"]) == 0
    assert int(s1["sbhash"], 16) == _fnv(sb)


def test_no_byte_behind_len_is_read_from_the_callers_buffer(tmp_path):
    doc = load_fixture("twitter.json")
    p = tmp_path / "twitter.json"
    p.write_bytes(doc)
    out = _run("guard", str(p))
    lens = [int(m) for m in re.findall(r"^len (\d+) ok", out, flags=re.M)]
    assert lens == [0, 1, 63, 64, 65, 4095, 4096, len(doc)] and ("count=55263" in out.splitlines()[-1])
