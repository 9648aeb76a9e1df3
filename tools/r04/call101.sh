#!/bin/bash
# round 4: the other counter groups for the mesh kernels' rows (TCC hit / miss, SQ cycle counts, VMEM / wait counts) -> gpurun_out/prof_r04_e2/summary/
cd "$GRAFT_REPO_ROOT"
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/prof_r04_e2; mkdir -p $OUT/summary
cd /tmp && export TMPDIR=/tmp
timeout 25 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_kernels" -- python "$ROOT/tools/run_kernels_once.py" mesh > "$OUT/trace_kernels.log" 2>&1 || echo "(trace failed)"
for grp in "TCC_HIT_sum TCC_MISS_sum" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  name=$(echo $grp | tr ' ' '+')
  timeout 25 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python "$ROOT/tools/run_kernels_once.py" mesh > "$OUT/pmc_$name.log" 2>&1 || echo "(pmc $grp failed)"
done
python "$ROOT/tools/summarize_counters.py" "$OUT" r04_e2
find "$OUT" -mindepth 1 -maxdepth 1 -type d ! -name summary -exec rm -rf {} +
