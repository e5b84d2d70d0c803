#!/bin/bash
# round 5, first GPU call: the new regression tests + per-kernel baseline of the configs[3] batch on this box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_zero_copy.py tests/test_gpu_c_caller.py tests/test_gpu_pipeline.py -x -q > gpurun_out/r5_t1.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5_t1.log
bash tools/prof_batch_r4.sh r5_base 2>&1 | tail -40
