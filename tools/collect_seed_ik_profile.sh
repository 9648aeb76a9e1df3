set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_seedik
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/seed_ik_once.py > $OUT/plain.log 2>&1; tail -1 $OUT/plain.log
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/tools/seed_ik_once.py > $OUT/trace.log 2>&1 || echo trace failed
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA; do
  timeout 70 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python $ROOT/tools/seed_ik_once.py > $OUT/pmc_$c.log 2>&1 || echo "$c failed"
done
ls $OUT
