# instruction counts and time of the fused launch by elimination of its cost passes (C2 shapes, 1024 trajectories)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT}
OUT=$ROOT/gpurun_out/r05_phase
mkdir -p $OUT
for v in "" "--no-self" "--no-scene" "--no-self --no-scene" "--no-sweep"; do
  n=$(echo "full$v" | tr -d ' ')
  python $ROOT/tools/r05/fused_variant.py $v --time >> $OUT/times.jsonl 2>> $OUT/err.log
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
    m=$(echo $c | tr ' ' '_')
    timeout 90 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$n/$m -- python $ROOT/tools/r05/fused_variant.py $v > $OUT/$n.$m.log 2>&1 || echo "$n $c failed"
  done
done
python - <<'PY'
import csv, glob, collections, os, json
out=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r05_phase"
res={}
for d in sorted(glob.glob(out+"/full*")):
    if not os.path.isdir(d): continue
    acc=collections.defaultdict(list)
    for p in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "rollout_trajectory_fused" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[os.path.basename(d)]={k: round(sum(v)/len(v)) for k,v in acc.items()}
json.dump(res, open(out+"/counters.json","w"), indent=1)
print(json.dumps(res, indent=1))
print(open(out+"/times.jsonl").read())
PY
