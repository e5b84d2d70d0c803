#!/bin/bash
# A/B of library variants on the HEADLINE (twitter.json x6801 = 4 GiB, bench.py's main section): tools/ab_headline.sh name [name ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "$@"; do
  if [ "$v" = cur ]; then unset SJMI_LIB; else export SJMI_LIB=$R/tools/variants/libsjmi_$v.so; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>&1 | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', 'value', d['value'], 'ms', r['avg_kernel_ms'], 'frac', r['frac'], 'settled', r['settled_frac'], 'cold', r['cold_frac'])"
done
