#!/bin/bash
# A/B of library variants / env settings on the configs[3] batch: "name|ENV=.. ENV=..|lib" triples; prints ms per batch and the main kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; lib=${rest#*|}
  [ "$lib" = "$rest" ] && lib=""
  [ -n "$lib" ] && export SJMI_LIB=$R/tools/variants/libsjmi_$lib.so || unset SJMI_LIB
  out=$(env $envs timeout 200 python bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 2 --warmup 1 --preheat 0 --batch-steps 20 --sample 200 --reps 64 2>/dev/null | tail -1)
  echo "$name: $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['extra']['batch_1m_docs']; print(b['ms_per_batch'], 'ms', b['value'], 'docs/s')")"
done
