"""The RCCL call path of the batched multi-GPU mode on the one GPU a test box has: `bench.py --gpus 1 --sharded` initialises
a real "nccl" process group of one rank, runs the sharded step -- sjmi_parse_batch_device on the (whole) shard, then the count
gather as an all_gather_into_tensor of device tensors -- and its documents/s must be the single-GPU figure of the same run
(the collective is 32 bytes: it may not cost anything measurable).  The protocol for world_size > 1 is covered on CPU by
tests/test_sharding_gloo.py; the 8-GPU curve is the driver's."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_sharded_branch_over_nccl_with_one_rank():
    env = dict(os.environ)
    env.pop("SJMI_BENCH_BACKEND", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sharded", "--steps", "10", "--warmup", "3",
                          "--docs", "400000"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["backend"].startswith("nccl")
    assert line["config"]["documents"] == 400000 and line["config"]["documents_per_rank"] == [400000]
    single = line["single_gpu_same_run"]["value"]
    # (the sharded step adds one 32-byte collective and two tiny torch ops to ~2.7 ms of kernels)
    assert abs(line["value"] - single) / single < 0.08, (line["value"], single)
