cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT}
OUT=$ROOT/gpurun_out/prof_sq
mkdir -p $OUT
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  n=$(echo $c | tr ' ' '_')
  timeout 60 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$n -- python $ROOT/tools/run_fused_once.py > $OUT/$n.log 2>&1 || echo "$c failed"
done
python - <<'PY'
import csv, glob, collections, os
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_sq"
acc=collections.defaultdict(list)
for p in glob.glob(out+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rollout_trajectory_fused" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY
