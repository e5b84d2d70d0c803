#!/bin/bash
# A/B comparison of library variants in ONE gpurun call (experiments only): steady-state kernel time percentiles
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for v in "$@"; do
  echo -n "$v: "
  SJMI_LIB=$R/tools/variants/$v.so timeout 60 python $R/tools/perlaunch.py 0 300 2>/dev/null | head -1 | tr " " "\n" | tail -100 | sort -n | sed -n "10p;50p;90p" | paste -sd" "
done
done
