REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/lob_trace -- python $REPO/tools/prof_lobpcg.py 22 > $REPO/gpurun_out/lob_trace.log 2>&1
cd $REPO
tail -2 gpurun_out/lob_trace.log
python - <<'PY'
import csv, glob
f = max(glob.glob("gpurun_out/lob_trace/**/*kernel_stats.csv", recursive=True))
for r in csv.DictReader(open(f)):
    print("%-90s calls %5s avg %9.1f us  tot %8.2f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
