ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_terms
mkdir -p $OUT
L=$ROOT/curobo_amd/lib
cp $L/libcurobo_hip.so $OUT/.orig.so
t() { echo -n "$1 | " >> $OUT/t.txt; env $2 python $ROOT/tools/r05/$3 >> $OUT/t.txt 2>> $OUT/err.log; }
#t "base generic coll TERMS" "CUROBO_HIP_FUSED_NO_SHAPES=1 CUROBO_HIP_FORCE_TERMS=1" "fused_variant.py --time"
#t "base generic trajopt" "CUROBO_HIP_FUSED_NO_SHAPES=1" "trajopt_variant.py"
#t "base shapes trajopt" "X=1" "trajopt_variant.py"
#t "base shapes trajopt 8 seeds" "X=1" "trajopt_variant.py --seeds 8"
cp $L/variants/libcurobo_hip_termsdense.so $L/libcurobo_hip.so
t "dense generic coll TERMS" "CUROBO_HIP_FUSED_NO_SHAPES=1 CUROBO_HIP_FORCE_TERMS=1" "fused_variant.py --time"
t "dense generic trajopt" "CUROBO_HIP_FUSED_NO_SHAPES=1" "trajopt_variant.py"
t "dense generic trajopt 8 seeds" "CUROBO_HIP_FUSED_NO_SHAPES=1" "trajopt_variant.py --seeds 8"
cp $OUT/.orig.so $L/libcurobo_hip.so; rm $OUT/.orig.so
cat $OUT/t.txt; grep -v amdgpu.ids $OUT/err.log | tail -5
