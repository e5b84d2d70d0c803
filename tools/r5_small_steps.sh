#!/bin/bash
# granule size of the plain pass (SJMI_BATCH_STEPS x 4 KiB) on small batches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for docs in 125000 250000 1000000; do
for st in 0 1 2; do
  out=$(SJMI_BATCH_STEPS=$st timeout 200 python bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 2 --warmup 1 --preheat 0 --batch-steps 40 --sample 200 --reps 64 --docs $docs 2>/dev/null | tail -1)
  echo "docs $docs steps $st: $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['extra']['batch_1m_docs']; print(b['ms_per_batch'], 'ms', b['value'], 'docs/s')")"
done
done
