ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_small_batch
mkdir -p $OUT
for s in 8 256; do
  python $ROOT/tools/r05/phase_stamps.py --seeds $s $PS_ARGS >> $OUT/stamps.txt 2>> $OUT/err2.log
  python $ROOT/tools/r05/phase_stamps.py --seeds $s --no-reorder $PS_ARGS >> $OUT/stamps.txt 2>> $OUT/err2.log
done
cat $OUT/stamps.txt; grep -v amdgpu.ids $OUT/err2.log | tail -5
