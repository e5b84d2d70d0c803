#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 2400 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_two_process.py -x -q > gpurun_out/r5_t5.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r5_t5.log
timeout 600 python tools/trip_rate.py 3000 neighbours=7 > gpurun_out/r5_trip7.json 2> gpurun_out/r5_trip7.err; echo "trip rc=$?"; tail -2 gpurun_out/r5_trip7.json
timeout 600 python tools/trip_rate.py 3000 rccl neighbours=7 > gpurun_out/r5_trip7r.json 2> gpurun_out/r5_trip7r.err; echo "trip rc=$?"; tail -2 gpurun_out/r5_trip7r.json
