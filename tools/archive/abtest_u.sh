#!/bin/bash
# A/B of library variants on the unescape path (experiments only)
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for v in "$@"; do
  echo -n "$v: "
  SJMI_LIB=$R/tools/variants/$v.so timeout 100 python $R/tools/unescape_prof.py 1024 5 2>/dev/null | grep "unescape total"
done
done
