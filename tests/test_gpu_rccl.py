"""The multi-GPU launch path of bench.py on the one GPU a test box has.

* `bench.py --gpus 1 --sharded` initialises a real "nccl" (= RCCL) process group of one rank and runs the sharded branch: stage 1
  of the rank's twitter.json documents + the count gather as an all_gather_into_tensor of device tensors, then configs[3] weak
  and strong (sjmi_parse_batch_device + the gather through sharding.sharded_step); with one rank weak == strong == the
  single-GPU figure of the same run (the collective is 32 bytes: it may not cost anything measurable).
* `python bench.py --gpus 2` on a box with ONE GPU must fail loudly (exit != 0), not measure one GPU and print n_gpus 1; so must
  a process group whose size is not --gpus.
* two gloo ranks sharing the one GPU (test hooks SJMI_BENCH_BACKEND=gloo, SJMI_BENCH_OVERSUBSCRIBE=1) run the FULL sharded
  path -- self-spawned by `python bench.py --gpus 2` -- and the line carries both ranks' counts.
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
