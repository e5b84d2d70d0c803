cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for f in 0 1 0 1; do
  out=gpurun_out/abl_s1_$f; rm -rf $out; mkdir -p $out
  SJMI_DBG_AFTER_WARMUP=$f timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python tools/batch_nocheck.py 1000000 10 > $out/run.log 2>&1
  python - $out <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]+'/t_kernel_trace.csv')) if 'k_stage1_batch' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print(sys.argv[1], "k_stage1_batch launches", len(d), "first two", d[:2], "last ten avg %.1f min %.1f" % (sum(d[-10:])/10, min(d[-10:])))
PY
done
