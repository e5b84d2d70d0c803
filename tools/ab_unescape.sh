#!/bin/bash
# A/B of library variants on the string pass of twitter x1024 (bench.py's unescape section): "name|ENV=..|lib"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; lib=${rest#*|}
  [ "$lib" = "$rest" ] && lib=""
  [ -n "$lib" ] && export SJMI_LIB=$R/tools/variants/libsjmi_$lib.so || unset SJMI_LIB
  out=$(env $envs timeout 200 python bench.py --no-cpu-baseline --sections unescape --skip-main-timing --steps 2 --warmup 1 --preheat 0 --reps 1024 2>/dev/null | tail -1)
  echo "$name: $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['extra']['unescape_twitter_x1024']; print(b['roofline']['avg_ms_per_call'], 'ms', b['roofline']['frac'])")"
done
