#!/bin/bash
# A/B of library variants (tools/variants/libsjmi_<name>.so; "cur" = the in-tree library) on the stage-1 headline kernel over
# twitter.json x1024: ms per launch and roofline fractions.  usage: tools/ab_stage1.sh name [name ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "$@"; do
  if [ "$v" = cur ]; then unset SJMI_LIB; else export SJMI_LIB=$R/tools/variants/libsjmi_$v.so; fi
  timeout 100 python bench.py --reps 1024 --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', 'ms', r['avg_kernel_ms'], 'frac', r['frac'], 'settled', r['settled_frac'], 'cold', r['cold_frac'])"
done
