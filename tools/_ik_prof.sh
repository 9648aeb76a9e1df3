cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ikp; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python $GRAFT_REPO_ROOT/tools/_ik_loop.py > $OUT/log.txt 2>&1
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per solve (60 solves): %.1f us" % (tot/60/1e3))
for r in rows[:22]:
    print("%-90s calls/solve %5.1f  avg %7.1f us  per solve %7.1f us" % (r["Name"][:90], int(r["Calls"])/60, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/60/1e3))
PY
rm -rf $OUT/raw
