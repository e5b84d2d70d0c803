#!/bin/bash
# A/B of the stage-1 expansion variants (tools/variants/libsjmi_<name>.so) in ONE call: the 1 M-document batch per kernel, twitter x1024 and
# the 4 GiB synthetic through bench.py.  usage: tools/ab_s1_sort.sh name [name ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
specs=""
for v in "$@"; do specs="$specs $v:$v"; done
tools/ab_kernels.sh batch $specs 2>&1 | grep -E "==|k_stage1_batch"
for v in "$@"; do
  export SJMI_LIB=$R/tools/variants/libsjmi_$v.so
  timeout 200 python bench.py --reps 1024 --no-cpu-baseline --sections synth --steps 100 --warmup 20 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$v', 'x1024 ms', r['avg_kernel_ms'], 'frac', r['frac'], 'settled', r['settled_frac'], 'cold', r['cold_frac'], '| synth', {k: v for k, v in c.items() if 'synth' in k and ('frac' in k or 'ms' in k)})"
done
