ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_terms2
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_fused_shapes.py tests/test_gpu_trajopt.py tests/test_gpu_fused.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
t() { echo -n "$1 | " >> $OUT/t.txt; env $2 python $ROOT/tools/r05/$3 >> $OUT/t.txt 2>> $OUT/err.log; }
t "shapes trajopt 1024" "X=1" "trajopt_variant.py"
t "shapes trajopt 32" "X=1" "trajopt_variant.py --seeds 8"
t "generic trajopt 1024" "CUROBO_HIP_FUSED_NO_SHAPES=1" "trajopt_variant.py"
t "shapes coll FORCE_TERMS" "CUROBO_HIP_FORCE_TERMS=1" "fused_variant.py --time"
t "shapes coll" "X=1" "fused_variant.py --time"
cat $OUT/t.txt
