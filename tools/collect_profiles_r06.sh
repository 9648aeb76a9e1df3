#!/bin/bash
# Run ON THE GPU BOX (through gpurun): everything profiles/r06_* is made from (round 6: + the ik config of run_kernels_once.py).
#   1. rocprofv3 --kernel-trace --stats of the default bench command and of the driver's command
#   2. rocprofv3 --kernel-trace --stats of tools/run_kernels_once.py (every kernel of C2 - C5, plain launches)
#   3. counter passes over tools/run_kernels_once.py, --pmc with --kernel-trace only, one group per run:
#      FETCH_SIZE | WRITE_SIZE (they do not fit one pass) | TCC hit / miss | SQ instruction counts | SQ cycle counts
#   4. tools/summarize_counters.py -> summary/<tag>_counters_by_kernel.{json,csv}: per (kernel, grid) launches,
#      mean duration, HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction), instruction counts,
#      VALU issue fraction
# Usage: bash tools/collect_profiles_r03.sh <tag> [configs...]   Output: gpurun_out/prof_<tag>/summary/
set -u
TAG=${1:-r06}
shift || true
CFGS=${*:-c2 c3 c4 c5 mesh trajopt ik}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT/summary"
cd /tmp && export TMPDIR=/tmp
if [ -z "${COUNTERS_ONLY:-}" ]; then
  echo "== kernel trace + stats: default bench command"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- \
    python "$ROOT/bench.py" --no-cpu-baseline --no-ik --no-configs > "$OUT/summary/${TAG}_bench_c2_under_kernel_trace.json" 2> "$OUT/trace_default.log" || echo "   (failed)"
  echo "== kernel trace + stats: the driver's command"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_driver" -- \
    python "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ik --no-configs > "$OUT/summary/${TAG}_bench_driver_cmd_under_kernel_trace.json" 2> "$OUT/trace_driver.log" || echo "   (failed)"
fi
echo "== kernel trace + stats: tools/run_kernels_once.py $CFGS"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_kernels" -- \
  python "$ROOT/tools/run_kernels_once.py" $CFGS > "$OUT/trace_kernels.log" 2>&1 || echo "   (failed)"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  name=$(echo $grp | tr ' ' '+')
  echo "== pmc $grp"
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- \
    python "$ROOT/tools/run_kernels_once.py" $CFGS > "$OUT/pmc_$name.log" 2>&1 || echo "   (failed or timed out: see pmc_$name.log)"
done
python "$ROOT/tools/summarize_counters.py" "$OUT" "$TAG"
ls -la "$OUT/summary"
# the raw traces stay on the box: only summary/ and the logs travel back (gpurun_out is capped at 64 MiB)
find "$OUT" -mindepth 1 -maxdepth 1 -type d ! -name summary -exec rm -rf {} +
