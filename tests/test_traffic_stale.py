"""bench.py quotes `roofline.traffic` from a tracked rocprofv3 summary (profiles/<round>/pmc_summary.json); `traffic_stale` must say
when the kernels' sources are no longer the ones the counters were collected from (tools/csrc_digest.py)."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _tree(tmp_path):
    root = tmp_path / "repo"
    os.makedirs(root / "simdjson-java_amd" / "csrc" / "host")
    os.makedirs(root / "include")
    (root / "simdjson-java_amd" / "csrc" / "a.hip").write_text("__global__ void k() {}\n")
    (root / "simdjson-java_amd" / "csrc" / "host" / "b.cpp").write_text("int f();\n")
    (root / "include" / "sjmi.h").write_text("int sjmi_x(void);\n")
    return str(root)


def test_digest_follows_the_kernel_sources_only(tmp_path):
    import csrc_digest as D
    root = _tree(tmp_path)
    d0 = D.csrc_digest(root)
    summary = {"_collected": {"csrc_digest": d0, "commit": "abc1234"}}
    assert D.traffic_stale(summary, root)[0] is False
    with open(os.path.join(root, "README.md"), "w") as f:  # not a kernel source: still valid
        f.write("x")
    assert D.traffic_stale(summary, root)[0] is False
    with open(os.path.join(root, "simdjson-java_amd", "csrc", "a.hip"), "a") as f:
        f.write("// changed\n")
    assert D.csrc_digest(root) != d0
    assert D.traffic_stale(summary, root)[0] is True
    shutil.rmtree(os.path.join(root, "simdjson-java_amd", "csrc", "host"))  # a removed file changes it too
    assert D.traffic_stale({"_collected": {"csrc_digest": D.csrc_digest(root)}}, root)[0] is False


def test_a_summary_without_a_stamp_is_stale_when_git_cannot_vouch_for_it(tmp_path):
    import csrc_digest as D
    root = _tree(tmp_path)  # (no .git there)
    assert D.traffic_stale({"_collected": {"commit": "ea01fd9"}}, root)[0] is True
    assert D.traffic_stale({}, root)[0] is True


def test_bench_line_carries_the_flag():
    """the roofline object of the bench line has `traffic_stale` whenever it has `traffic`"""
    import bench
    r = bench.roofline(1000, 1.0, 1000, "k", 1, traffic=1234)
    assert r["traffic"] == 1234 and isinstance(r["traffic_stale"], bool) and r["traffic_source"]
    assert "traffic_stale" not in bench.roofline(1000, 1.0, 1000, "k", 1)


def test_the_memory_system_yardstick_is_optional(tmp_path, monkeypatch):
    """bench.memory_system_rate(): the do-nothing kernel's figures beside the roofline are a yardstick -- without the binary
    (tools/ubench/load_pattern, built by __graft_entry__.build()) the line simply does not carry them."""
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.memory_system_rate() is None
