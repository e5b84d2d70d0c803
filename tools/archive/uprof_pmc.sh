cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $R/gpurun_out/upmc_$tag -o p -- python $R/tools/unescape_prof.py 1024 2 > $R/gpurun_out/upmc_$tag.log 2>&1
done
python - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$R/gpurun_out/upmc_SQ_WAVES/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_str" in k:
            a=agg[(k.split("(")[0][-14:], r["Counter_Name"])]; a[0]+=float(r["Counter_Value"]); a[1]+=1
    for k,v in sorted(agg.items()): print(k, round(v[0]/v[1]))
PY
