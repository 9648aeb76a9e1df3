# time_variants.sh name1 name2 ...: exclusive C2 launch time of each variant library (curobo_amd/lib/variants/), on the GPU box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_variants
mkdir -p $OUT
cp $ROOT/curobo_amd/lib/libcurobo_hip.so $OUT/.orig.so
for n in "$@"; do
  cp $ROOT/curobo_amd/lib/variants/libcurobo_hip_$n.so $ROOT/curobo_amd/lib/libcurobo_hip.so
  for rep in 1 2; do
    echo -n "$n: " >> $OUT/times.txt
    python $ROOT/tools/r05/fused_variant.py --time $EXTRA_ARGS >> $OUT/times.txt 2>> $OUT/err.log
  done
done
cp $OUT/.orig.so $ROOT/curobo_amd/lib/libcurobo_hip.so; rm $OUT/.orig.so
cat $OUT/times.txt
