# k_coop_walk on the bench's 1M-document batch under the ablation switches of coop_walk.hip (SJMI_COOP_ABLATE bits:
# 1 = no primitive parsing, 2 = no level loop, 4 = no tape stores): which part of the step costs what
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in 0 1 2 4 7; do
  rm -rf gpurun_out/abl_$a
  SJMI_COOP_ABLATE=$a rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$a -o t -- python bench.py --no-cpu-baseline --sections batch --skip-main-timing --steps 3 --warmup 1 --preheat 0 --batch-steps 3 > gpurun_out/abl_$a.log 2>&1
  python - $a <<'PY'
import csv, sys
a = sys.argv[1]
for r in csv.DictReader(open('gpurun_out/abl_%s/t_kernel_stats.csv' % a)):
    if 'k_coop_walk' in r['Name'] or 'k_str_measure' in r['Name'] or 'k_str_write' in r['Name']:
        print("ablate", a, r['Name'][:40], r['Calls'], "avg %.1f us" % (float(r['AverageNs']) / 1e3))
PY
done
