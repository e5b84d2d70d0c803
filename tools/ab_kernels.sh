#!/bin/bash
# A/B of library variants on the GPU box, per KERNEL (rocprofv3 --kernel-trace --stats): tools/ab_kernels.sh <workload> name[:lib] ...
#   workload = batch   the configs[3] batch through the optimistic pipeline, no checks (tools/batch_nocheck.py)
#            = unescape | x1024   bench.py's section of that name (twitter x1024)
# lib = a library built by tools/build_variant.sh (tools/variants/libsjmi_<lib>.so); none: the tree's libsjmi.so
R=${GRAFT_REPO_ROOT:-/root/repo}
wl=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $R
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec#*:}
  [ "$lib" = "$spec" ] && lib=""
  [ -n "$lib" ] && export SJMI_LIB=$R/tools/variants/libsjmi_$lib.so || unset SJMI_LIB
  out=gpurun_out/ab_${wl}_$name
  rm -rf $out; mkdir -p $out
  case $wl in
    batch) cmd="python tools/batch_nocheck.py 1000000 10";;
    *) cmd="python bench.py --no-cpu-baseline --sections $wl --skip-main-timing --steps 20 --warmup 2 --preheat 0 --reps 64";;
  esac
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- $cmd > $out/run.log 2>&1
  echo "== $name (rc=$?) $(grep -h 'ms per step' $out/run.log)"
  python - $out <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1] + '/t_kernel_stats.csv'))]
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:8]:
    n = r['Name'].replace('void ', '').replace('sjmi::', '').split('(')[0]
    if n.startswith('at::') or 'elementwise' in n or 'Cijk' in n or 'rocclr' in n: continue
    print("   %-40s calls %5s avg %9.1f us  min %9.1f" % (n[:40], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
