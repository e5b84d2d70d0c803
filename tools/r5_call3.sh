#!/bin/bash
# round 5: the restructured batch pipeline -- parity of everything that goes through it, then its per-kernel profile
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_walk.py tests/test_gpu_batch.py tests/test_gpu_coop_walk.py tests/test_gpu_fullscale.py tests/test_gpu_rccl.py -x -q > gpurun_out/r5_t3.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r5_t3.log
bash tools/prof_batch_r4.sh r5_pipe 2>&1 | tail -30
