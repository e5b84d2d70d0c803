"""The multi-GPU launch path of bench.py on the one GPU a test box has.

* `bench.py --gpus 1 --sharded` initialises a real "nccl" (= RCCL) process group of one rank and runs the sharded branch: stage 1
  of the rank's twitter.json documents + the count gather as an all_gather_into_tensor of device tensors, then configs[3] weak
  and strong (sjmi_parse_batch_device + the gather through sharding.sharded_step); with one rank weak == strong == the
  single-GPU figure of the same run (the collective is 32 bytes: it may not cost anything measurable).
* `python bench.py --gpus 2` on a box with ONE GPU must fail loudly (exit != 0), not measure one GPU and print n_gpus 1; so must
  a process group whose size is not --gpus.
* two gloo ranks sharing the one GPU (test hooks SJMI_BENCH_BACKEND=gloo, SJMI_BENCH_OVERSUBSCRIBE=1) run the FULL sharded
  path -- self-spawned by `python bench.py --gpus 2` -- and the line carries both ranks' counts.
* EIGHT gloo ranks on the one GPU through `python bench.py --gpus 8` (round 5: the launch the driver makes on an 8-GPU node, as far
  as one GPU allows: eight processes, eight engine contexts, eight sets of persistent kernels contending, the gather over 8 ranks).
* the SCALE-style N = 1 line (`--sharded`, real RCCL group of one) against the BENCH-style line of the same box, same workload:
  the headline value must agree within 5 % -- the driver computes scaling efficiency as value(N) / (N x value(1)) across the two.
The protocol for world_size > 1 is also covered on CPU by tests/test_sharding_gloo.py; the 8-GPU curve is the driver's."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _env(**more):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SJMI_BENCH_BACKEND", "SJMI_BENCH_OVERSUBSCRIBE"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(more)
    return env


def _line(out):
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_sharded_branch_over_nccl_with_one_rank():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sharded", "--steps", "10", "--warmup", "3",
                          "--preheat", "5", "--reps", "1024", "--docs", "200000", "--sample", "500"],
                         capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _line(out)
    assert line["n_gpus"] == 1 and line["backend"].startswith("nccl") and line["config"]["rccl_world_size"] == 1
    assert line["unit"] == "GB/s" and line["scaling"] == "weak" and line["config"]["documents_per_rank"] == [1024]
    b = line["batched"]
    assert b["weak"]["documents_per_rank"] == [200000] and b["strong"]["documents_per_rank"] == [200000]
    assert b["weak"]["rccl_world_size"] == 1 and b["oracle_checked_documents_per_rank"] == 1000
    single = b["single_gpu_same_run"]["value"]
    # (the sharded step adds one 32-byte collective and two tiny torch ops to ~1.3 ms of kernels)
    assert abs(b["weak"]["value"] - single) / single < 0.10, (b["weak"]["value"], single)


def test_more_ranks_than_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert out.returncode != 0 and "refusing to run" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # a rank that finds WORLD_SIZE != --gpus refuses as well
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert out.returncode != 0 and "refusing to run" in out.stderr


def test_two_gloo_ranks_share_the_gpu_through_the_self_spawned_launch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--preheat", "2",
                          "--reps", "256", "--docs", "100000", "--batch-steps", "4", "--sample", "400"],
                         capture_output=True, text=True, timeout=900,
                         env=_env(SJMI_BENCH_BACKEND="gloo", SJMI_BENCH_OVERSUBSCRIBE="1"), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _line(out)
    assert line["n_gpus"] == 2 and line["config"]["rccl_world_size"] == 2 and line["backend"] == "gloo"
    assert line["config"]["documents_per_rank"] == [256, 256]
    b = line["batched"]
    assert b["weak"]["documents_per_rank"] == [100000, 100000] and sum(b["strong"]["documents_per_rank"]) == 100000
    assert all(40000 < x < 60000 for x in b["strong"]["documents_per_rank"])  # byte-balanced halves


def test_eight_gloo_ranks_share_the_gpu_through_the_self_spawned_launch():
    """What the driver's 8-GPU run executes, with everything but the GPUs real: `python bench.py --gpus 8` self-spawns eight ranks under
    torch.distributed.run (here over gloo, all on the one GPU); every rank generates only its own documents (in slices through one
    re-used pinned buffer), checks a sample of its shard against the oracle, and the line carries all eight ranks' counts."""
    docs = 40000
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--preheat", "1",
                          "--reps", "64", "--docs", str(docs), "--batch-steps", "3", "--sample", "800"],
                         capture_output=True, text=True, timeout=1500,
                         env=_env(SJMI_BENCH_BACKEND="gloo", SJMI_BENCH_OVERSUBSCRIBE="1"), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _line(out)
    assert line["n_gpus"] == 8 and line["config"]["rccl_world_size"] == 8 and line["backend"] == "gloo"
    assert line["config"]["documents_per_rank"] == [64] * 8 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["roofline"]["all_ranks_peak"] == 8 * 8000.0
    b = line["batched"]
    assert b["weak"]["documents_per_rank"] == [docs] * 8 and b["weak"]["documents"] == 8 * docs and b["weak"]["rccl_world_size"] == 8
    assert sum(b["strong"]["documents_per_rank"]) == docs and b["strong"]["rccl_world_size"] == 8
    assert all(docs / 8 * 0.8 < x < docs / 8 * 1.2 for x in b["strong"]["documents_per_rank"])  # byte-balanced eighths
    assert sum(b["weak"]["structurals_per_rank"]) > 8 * docs * 150 and b["oracle_checked_documents_per_rank"] == 200
    assert line["config"]["batched_documents"] == docs


def test_sharded_n1_line_agrees_with_the_bench_line():
    """value(N = 1) of the SCALE protocol (sharded branch over a real RCCL group of one: stage 1 + the count gather per step) against
    the BENCH line's value on the same box and workload (twitter.json x 6801 = 4 GiB): within 5 %."""
    common = ["--steps", "20", "--warmup", "5", "--preheat", "40"]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-extras", "--no-cpu-baseline"] + common,
                       capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sharded", "--docs", "50000", "--sample", "200",
                        "--batch-steps", "3"] + common, capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert b.returncode == 0, b.stderr[-2000:]
    la, lb = _line(a), _line(b)
    assert la["metric"] == lb["metric"] and la["unit"] == lb["unit"] == "GB/s"
    assert la["config"]["bytes_per_gpu"] == lb["config"]["bytes_per_gpu"] == 4294933515
    assert abs(lb["value"] - la["value"]) / la["value"] < 0.05, (la["value"], lb["value"])
