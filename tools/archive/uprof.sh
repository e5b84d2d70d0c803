cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/uprof -o t -- python $R/tools/unescape_prof.py > $R/gpurun_out/uprof.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/uprof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sjmi" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
