#!/bin/bash
# round 4: the mesh kernels' rows of the counter file after the select-kernel rewrite: kernel trace + three counter passes
# (FETCH_SIZE | WRITE_SIZE | SQ instruction counts) over tools/run_kernels_once.py mesh  ->  gpurun_out/prof_r04_e/summary/
cd "$GRAFT_REPO_ROOT"
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/prof_r04_e; mkdir -p $OUT/summary
cd /tmp && export TMPDIR=/tmp
timeout 25 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_kernels" -- python "$ROOT/tools/run_kernels_once.py" mesh > "$OUT/trace_kernels.log" 2>&1 || echo "(trace failed)"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"; do
  name=$(echo $grp | tr ' ' '+')
  timeout 25 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python "$ROOT/tools/run_kernels_once.py" mesh > "$OUT/pmc_$name.log" 2>&1 || echo "(pmc $grp failed)"
done
python "$ROOT/tools/summarize_counters.py" "$OUT" r04_e
find "$OUT" -mindepth 1 -maxdepth 1 -type d ! -name summary -exec rm -rf {} +
grep -h "select\|walk" $OUT/summary/r04_e_counters_by_kernel.csv | cut -c1-300
