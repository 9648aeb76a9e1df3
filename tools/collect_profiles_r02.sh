#!/bin/bash
# Run ON THE GPU BOX (through gpurun): everything profiles/r02_* is made from.
#   1. tools/collect_profiles.sh <tag>: rocprofv3 --kernel-trace --stats of the default bench command, the HBM
#      PMC passes of the fused kernel (own runs), a plain bench line, the per-kernel / per-shape summaries
#   2. rocprofv3 --kernel-trace --stats of the secondary objects, one run each (bench.py --only c3 | c4 | c5)
#   3. SQ instruction counters of the fused kernel at the seed state (own runs) -> <tag>_fused_sq_counters.json
# Usage: bash tools/collect_profiles_r02.sh <tag>      Output: gpurun_out/prof_<tag>/summary/
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT/summary"
bash "$ROOT/tools/collect_profiles.sh" "$TAG"
cd /tmp && export TMPDIR=/tmp
for cfgname in c3 c4 c5; do
  echo "== kernel trace + stats, bench.py --only $cfgname"
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$cfgname" -- \
    python "$ROOT/bench.py" --no-cpu-baseline --only $cfgname --steps 20 --warmup 5 > "$OUT/summary/${TAG}_bench_${cfgname}_under_kernel_trace.json" 2> "$OUT/trace_$cfgname.log" \
    || echo "   (failed or timed out)"
  f=$(find "$OUT/trace_$cfgname" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/summary/${TAG}_kernel_stats_bench_${cfgname}.csv"
done
echo "== SQ counters of the fused kernel (seed state, 1024 trajectories per launch)"
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  n=$(echo $c | tr ' ' '_')
  timeout 60 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/sq_$n" -- python "$ROOT/tools/run_fused_once.py" > "$OUT/sq_$n.log" 2>&1 || echo "   $c failed"
done
OUT="$OUT" TAG="$TAG" python - <<'PY'
import csv, glob, collections, json, os
out, tag = os.environ["OUT"], os.environ["TAG"]
acc, kern = collections.defaultdict(list), None
for p in glob.glob(out + "/sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rollout_trajectory_fused" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kern = r["Kernel_Name"]
if acc:
    rec = {"command": "rocprofv3 --pmc <two counters> --kernel-trace --output-format csv -- python tools/run_fused_once.py "
                      "(tools/collect_profiles_r02.sh; seed-state C2 workload, 1024 trajectories per launch, per launch averages)",
           "kernel": kern.split("(")[0].replace("void curobo_hip::", "")}
    for k, v in sorted(acc.items()):
        rec[k] = round(sum(v) / len(v), 1)
        rec[k + "_launches"] = len(v)
    B = 1024
    rec["per_trajectory"] = {"valu_wave_instructions": round(rec.get("SQ_INSTS_VALU", 0) / B), "salu": round(rec.get("SQ_INSTS_SALU", 0) / B),
                             "lds": round(rec.get("SQ_INSTS_LDS", 0) / B)}
    if rec.get("SQ_WAVE_CYCLES"):
        rec["valu_active_share_of_wave_cycles"] = round(rec.get("SQ_ACTIVE_INST_VALU", 0) / rec["SQ_WAVE_CYCLES"], 4)
    json.dump(rec, open(f"{out}/summary/{tag}_fused_sq_counters.json", "w"), indent=1)
    print(json.dumps(rec["per_trajectory"]))
PY
ls -la "$OUT/summary"
# the raw traces stay on the box: only summary/ and the logs travel back (gpurun_out is capped at 64 MiB)
find "$OUT" -mindepth 1 -maxdepth 1 -type d ! -name summary -exec rm -rf {} +
