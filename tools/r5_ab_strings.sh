#!/bin/bash
# round 5: parity + A/B of k_strings variants (tools/variants/libsjmi_<name>.so); args: variant names
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "$@"; do
  if [ "$v" != "base" ]; then export SJMI_LIB=$R/tools/variants/libsjmi_$v.so; else unset SJMI_LIB; fi
  timeout 900 python -m pytest tests/test_gpu_unescape.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py -x -q > gpurun_out/r5_ab_$v.log 2>&1
  echo "$v tests rc=$? $(tail -1 gpurun_out/r5_ab_$v.log)"
done
unset SJMI_LIB
specs=""; uspecs=""
for v in "$@"; do if [ "$v" = "base" ]; then specs="$specs base||"; else specs="$specs $v||$v"; fi; done
bash tools/ab_batch.sh $specs
bash tools/ab_unescape.sh $specs
bash tools/ab_batch.sh $specs
