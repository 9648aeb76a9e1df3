#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + stats of the default bench command,
# then the PMC passes for HBM traffic (counters in their own runs, kernel trace only), and a summary
# grouped by kernel and launch shape.  Usage: bash tools/collect_profiles.sh <tag>
# Output: gpurun_out/prof_<tag>/ (raw) and gpurun_out/prof_<tag>/summary/ (small files for profiles/).
# Counter passes run the single-stream variants of the command (--shards 1; --seeds 64 for the
# 256-trajectory launch shape of the 4-shard default): rocprofv3 counter collection aborted on the
# multi-stream graph.  Every profiler run is bounded by `timeout`.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT/summary"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-ik --steps 100 --warmup 20"
if [ -z "${PMC_ONLY:-}" ]; then
echo "== kernel trace + stats (default command)"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_under_trace.json" 2> "$OUT/trace.log"
fi
for shape in 1024 256; do
  seeds=$((shape / 4))
  # FETCH_SIZE and WRITE_SIZE do not fit in one pass ("exceeds the capabilities of the hardware")
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    if [ -n "${PMC_ONLY:-}" ] && [ "$grp" = "TCC_HIT_sum TCC_MISS_sum" ]; then continue; fi
    name=$(echo $grp | cut -d' ' -f1)
    echo "== pmc $grp, $shape trajectories per launch"
    timeout 70 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_${shape}_$name" -- \
      python $ROOT/bench.py --no-cpu-baseline --no-ik --steps 20 --warmup 10 --shards 1 --seeds $seeds > /dev/null 2> "$OUT/pmc_${shape}_$name.log" \
      || echo "   (failed or timed out)"
  done
done
if [ -z "${PMC_ONLY:-}" ]; then
echo "== plain bench (no profiler)"
timeout 300 python "$ROOT/bench.py" > "$OUT/summary/${TAG}_bench_c2.json" 2> "$OUT/bench.log"
fi
python "$ROOT/tools/summarize_profiles.py" "$OUT" "$TAG"
ls -la "$OUT/summary"
