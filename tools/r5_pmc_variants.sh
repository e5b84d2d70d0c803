#!/bin/bash
# SQ counters of the batch pipeline for library variants: args = variant names (base = the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "$@"; do
  if [ "$v" != "base" ]; then export SJMI_LIB=$R/tools/variants/libsjmi_$v.so; else unset SJMI_LIB; fi
  echo "== $v"
  bash tools/pmc_batch.sh r5_$v 2>&1 | grep -v "^pmc rc"
done
