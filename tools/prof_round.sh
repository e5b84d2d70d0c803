#!/bin/bash
# A round's rocprofv3 collection (run on the GPU box via gpurun): tools/prof_round.sh r5
#   1. --kernel-trace --stats of the DEFAULT bench command (python bench.py --no-cpu-baseline): every kernel the bench line quotes
#   2. FETCH_SIZE and WRITE_SIZE, separate passes (never combined with trace domains), one bench section per pass so that
#      the k_stage1 dispatches of the three workloads cannot be confused; SQ counters for the unescape and batch kernels
#   3. the calibration kernels (tools/pmc_calibrate.py) under the same two counters
# Output: gpurun_out/prof_<round>/... ; tools/summarize_prof_round.py turns it into gpurun_out/profiles_<round>/, tools/install_prof.py
# <round> (in the build container) into the tracked profiles/<round>/.
R=${GRAFT_REPO_ROOT:-/root/repo}
RD=${1:-r6}
out=$R/gpurun_out/prof_$RD
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/trace.log 2>&1
echo "trace rc=$?"; tail -c 400 $out/trace.log | tr '\n' ' ' | cut -c1-300; echo
P="--steps 3 --warmup 1 --preheat 0 --batch-steps 2"
for sec in main x1024 unescape synth batch; do
  if [ $sec = main ]; then S="--no-extras"; else S="--sections $sec --skip-main-timing --batch-accepted-only"; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/pmc_${sec}_$ctr -o p -- $B $P $S > $out/pmc_${sec}_$ctr.log 2>&1
    echo "pmc $sec $ctr rc=$?"
  done
done
for sec in unescape batch; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/pmc_${sec}_SQ -o p -- $B $P --sections $sec --skip-main-timing --batch-accepted-only > $out/pmc_${sec}_SQ.log 2>&1
  echo "pmc $sec SQ rc=$?"
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $out/cal_$ctr -o p -- python $R/tools/pmc_calibrate.py > $out/cal_$ctr.log 2>&1
  echo "cal $ctr rc=$?"
done
python $R/tools/summarize_prof_round.py $out $R/gpurun_out/profiles_$RD
