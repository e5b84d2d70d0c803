#!/bin/bash
# PC sampling (rocprofv3 --pc-sampling-beta-enabled, host_trap) of the configs[3] batch: where the waves of a kernel are.
# usage (GPU box): tools/pcsamp_batch.sh <tag> [interval_us] [method]; output gpurun_out/pcsamp_<tag>/hist.txt (aggregated on the box)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-x}; iv=${2:-200}; method=${3:-host_trap}
out=$R/gpurun_out/pcsamp_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
unit=time; [ $method = stochastic ] && unit=cycles
B="python $R/bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 2 --warmup 1 --preheat 0 --batch-steps 6 --sample 200 --reps 64"
timeout 240 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $method --pc-sampling-interval $iv \
   --output-format csv -d /tmp/pcs -o s -- $B > $out/run.log 2>&1
echo "rc=$?"
ls -la /tmp/pcs | head -20
python $R/tools/pcsamp_hist.py /tmp/pcs $out
