# small_batch_probe.sh: exclusive fused C2 launch at small batches, workgroup sizes 512 (default shape) / 768 / 1024 (run-time shapes)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_small_batch
mkdir -p $OUT
for s in 8 32 128; do
  echo -n "seeds $s threads default: " >> $OUT/times.txt
  python $ROOT/tools/r05/fused_variant.py --time --seeds $s >> $OUT/times.txt 2>> $OUT/err.log
  for t in 768 1024; do
    echo -n "seeds $s threads $t: " >> $OUT/times.txt
    CUROBO_HIP_FUSED_THREADS=$t CUROBO_HIP_JIT_SHAPES=1 python $ROOT/tools/r05/fused_variant.py --time --seeds $s >> $OUT/times.txt 2>> $OUT/err.log
  done
done
cat $OUT/times.txt; tail -5 $OUT/err.log
