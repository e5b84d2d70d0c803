"""The reference timed beside the GPU: java/bench/org/simdjson/RefStage1Bench.java + bench.py's
hook.  No JVM in this image, so, like tests/test_java_shim.py, everything is verified as far as text and stubs allow:
* the harness is in package org.simdjson and every package-private member of the reference it calls exists there with the
  arity it is called with (needs /root/reference: build container only);
* the javac / java command lines bench.py would run are well formed (all of src/main + the harness, the incubator module,
  -Dorg.simdjson.species=512, the class name and its four arguments);
* with a STUB JDK on PATH (two shell scripts) the hook runs end to end and the bench line's cpu_baseline upgrades itself to
  kind "reference" with the port kept beside it; without a JDK, or with one that is too old, it stays "port" and says why."""
import json
import os
import re
import stat
import sys

import pytest

from tests.conftest import ROOT

sys.path.insert(0, ROOT)
REF_MAIN = "/root/reference/src/main/java"
HARNESS = os.path.join(ROOT, "java", "bench", "org", "simdjson", "RefStage1Bench.java")


def _strip(java):
    java = re.sub(r"/\*.*?\*/", " ", java, flags=re.S)
    java = re.sub(r"//[^\n]*", " ", java)
    java = re.sub(r"'(?:\\.|[^'\\])'", "' '", java)  # (char literals first: '"' would open a string)
    return re.sub(r'"(?:\\.|[^"\\])*"', '""', java)


def test_harness_source_is_well_formed():
    src = open(HARNESS).read()
    code = _strip(src)
    assert re.search(r"^\s*package org\.simdjson;", code, flags=re.M), "package-private access needs package org.simdjson"
    assert code.count("{") == code.count("}") and code.count("(") == code.count(")")
    assert "public static void main(String[] args)" in code and "public final class RefStage1Bench" in code
    # the two passes of SimdJsonParser.stage1, in its order, on the same buffer and length
    body = code[code.index("stage1 = t ->"):]
    v, i = body.index("Utf8Validator.validate(mine, len)"), body.index("indexer.index(mine, len)")
    assert body.index("bits.reset()") < v < i
    for key in ("stage1_gb_per_s_one_thread", "stage1_gb_per_s_all_threads", "parse_per_s_one_thread", "parse_per_s_all_threads",
                "slice_bytes", "vector_bits"):
        assert key in src, key


@pytest.mark.skipif(not os.path.isdir(REF_MAIN), reason="the reference sources only exist in the build container")
def test_every_reference_member_the_harness_calls_exists():
    ref = {f[:-5]: _strip(open(os.path.join(REF_MAIN, "org", "simdjson", f)).read())
           for f in os.listdir(os.path.join(REF_MAIN, "org", "simdjson")) if f.endswith(".java")}

    def has(cls, pattern):
        assert re.search(pattern, ref[cls]), "%s: %s not found in the reference" % (cls, pattern)

    has("Utf8Validator", r"static\s+void\s+validate\s*\(\s*byte\[\]\s+\w+\s*,\s*int\s+\w+\s*\)")      # Utf8Validator.java:54
    has("StructuralIndexer", r"StructuralIndexer\s*\(\s*BitIndexes\s+\w+\s*\)")                        # :38
    has("StructuralIndexer", r"void\s+index\s*\(\s*byte\[\]\s+\w+\s*,\s*int\s+\w+\s*\)")
    has("BitIndexes", r"BitIndexes\s*\(\s*int\s+\w+\s*\)")
    has("BitIndexes", r"void\s+reset\s*\(\s*\)")
    has("BitIndexes", r"boolean\s+hasNext\s*\(\s*\)")
    has("BitIndexes", r"int\s+getAndAdvance\s*\(\s*\)")
    has("VectorUtils", r"static\s+final\s+VectorSpecies<Byte>\s+BYTE_SPECIES")
    has("SimdJsonParser", r"public\s+SimdJsonParser\s*\(\s*\)")
    has("SimdJsonParser", r"public\s+JsonValue\s+parse\s*\(\s*byte\[\]\s+\w+\s*,\s*int\s+\w+\s*\)")
    # none of them is private (the harness sits in the same package, not inside the classes)
    assert not re.search(r"private\s+static\s+void\s+validate", ref["Utf8Validator"])
    assert not re.search(r"private\s+void\s+index\s*\(", ref["StructuralIndexer"])
    # what it times is the body of SimdJsonParser.stage1 (SimdJsonParser.java:55-58)
    m = re.search(r"private\s+void\s+stage1\s*\([^)]*\)\s*\{(.*?)\}", ref["SimdJsonParser"], flags=re.S)
    assert m and "Utf8Validator.validate(buffer, length)" in m.group(1) and "indexer.index(buffer, length)" in m.group(1)
    # the species property the command line sets is the one VectorUtils reads
    assert '"org.simdjson.species"' in open(os.path.join(REF_MAIN, "org", "simdjson", "VectorUtils.java")).read()


def test_command_lines_are_well_formed(tmp_path):
    import bench
    src = tmp_path / "src" / "org" / "simdjson"
    src.mkdir(parents=True)
    for name in ("A.java", "B.java", "notes.txt"):
        (src / name).write_text("// stub\n")
    javac, java = bench.reference_jvm_commands(str(tmp_path / "src"), str(tmp_path / "out"), str(tmp_path / "doc.json"), 7, 4.0)
    assert os.path.basename(javac[0]) == "javac" and javac[1:3] == ["--add-modules", "jdk.incubator.vector"]
    assert javac[javac.index("-d") + 1] == str(tmp_path / "out")
    sources = [a for a in javac if a.endswith(".java")]
    assert sources[-1] == HARNESS and sorted(os.path.basename(a) for a in sources[:-1]) == ["A.java", "B.java"]
    assert not any(a.endswith(".txt") for a in javac)
    assert os.path.basename(java[0]) == "java" and java[1:3] == ["--add-modules", "jdk.incubator.vector"]
    assert "-Dorg.simdjson.species=512" in java and java[java.index("-cp") + 1] == str(tmp_path / "out")
    k = java.index("org.simdjson.RefStage1Bench")
    assert java[k + 1:] == [str(tmp_path / "doc.json"), "7", "4.0", str(32 << 20)]


def _stub_jdk(dirpath, version, line):
    dirpath.mkdir()
    for name, body in (("javac", 'if [ "$1" = "-version" ]; then echo "javac %s" >&2; exit 0; fi\nexit 0\n' % version),
                       ("java", 'if [ "$1" = "-version" ]; then echo \'openjdk version "%s" 2025-03-18\' >&2; exit 0; fi\n'
                                "echo 'WARNING: Using incubator modules: jdk.incubator.vector'\necho '%s'\n" % (version, line))):
        p = dirpath / name
        p.write_text("#!/bin/sh\n" + body)
        p.chmod(p.stat().st_mode | stat.S_IEXEC)


def test_hook_upgrades_the_baseline_with_a_stub_jdk(tmp_path, monkeypatch):
    import bench
    ref = tmp_path / "ref" / "org" / "simdjson"
    ref.mkdir(parents=True)
    (ref / "SimdJsonParser.java").write_text("package org.simdjson;\n")
    line = json.dumps({"harness": "RefStage1Bench", "species": "512", "vector_bits": 512, "java": "24.0.1", "threads": 4,
                       "slice_bytes": 33470295, "slice_structurals": 2928939, "seconds": 5.0, "stage1_gb_per_s_one_thread": 3.21,
                       "stage1_gb_per_s_all_threads": 11.5, "parse_per_s_one_thread": 1800.0, "parse_per_s_all_threads": 6900.0})
    _stub_jdk(tmp_path / "jdk24", "24.0.1", line)
    monkeypatch.setenv("PATH", str(tmp_path / "jdk24") + os.pathsep + os.environ["PATH"])
    monkeypatch.setattr(bench, "REFERENCE_SRC_CANDIDATES", (str(tmp_path / "ref"),))
    assert bench.java_major(str(tmp_path / "jdk24" / "java")) == 24 and bench.java_major(str(tmp_path / "jdk24" / "javac")) == 24
    res, why = bench.reference_jvm_baseline(b'{"a":1}', seconds=0.1)
    assert why == "ok" and res["stage1_gb_per_s_all_threads"] == 11.5 and res["cores"] >= 1
    # an old JDK, or no reference sources: the port stays, with the reason
    _stub_jdk(tmp_path / "jdk17", "17.0.9", line)
    monkeypatch.setenv("PATH", str(tmp_path / "jdk17") + os.pathsep + os.environ["PATH"])
    res, why = bench.reference_jvm_baseline(b"{}", seconds=0.1)
    assert res is None and "17 < 24" in why
    monkeypatch.setenv("PATH", str(tmp_path / "jdk24") + os.pathsep + os.environ["PATH"])
    monkeypatch.setattr(bench, "REFERENCE_SRC_CANDIDATES", (str(tmp_path / "nowhere"),))
    res, why = bench.reference_jvm_baseline(b"{}", seconds=0.1)
    assert res is None and "sources are not on this box" in why


def test_cpu_baseline_line_says_reference_when_the_hook_ran(tmp_path, monkeypatch):
    """bench.cpu_baseline: kind 'reference' + the port beside it when the harness ran; kind 'port' + the reason otherwise."""
    import bench
    from tests.conftest import load_fixture
    doc = load_fixture("github_events.json")
    fake = {"stage1_gb_per_s_all_threads": 12.25, "stage1_gb_per_s_one_thread": 3.5, "parse_per_s_all_threads": 7000.0,
            "parse_per_s_one_thread": 1900.0, "cores": 4, "vector_bits": 512, "java": "24", "slice_bytes": 1, "seconds": 5.0, "command": "java ..."}
    monkeypatch.setattr(bench, "reference_jvm_baseline", lambda d, seconds=5.0: (fake, "ok"))
    got = bench.cpu_baseline(doc, seconds=0.3)
    assert got["kind"] == "reference" and got["value"] == 12.25 and got["port"]["kind"] == "port" and got["port"]["value"] > 0
    monkeypatch.setattr(bench, "reference_jvm_baseline", lambda d, seconds=5.0: (None, "no JDK on PATH"))
    got = bench.cpu_baseline(doc, seconds=0.3)
    assert got["kind"] == "port" and got["reference_jvm"] == "not run: no JDK on PATH"
