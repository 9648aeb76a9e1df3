ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_terms5
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_trajopt.py tests/test_fused_shapes.py tests/test_gpu_fused.py tests/test_gpu_planner.py tests/test_gpu_multiframe_ik.py tests/test_gpu_randomised_sweeps.py tests/test_gpu_api.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
t() { echo -n "$1 | " >> $OUT/t.txt; env $2 python $ROOT/tools/r05/$3 >> $OUT/t.txt 2>> $OUT/err.log; }
t "trajopt 1024" "X=1" "trajopt_variant.py"
t "trajopt 1024" "X=1" "trajopt_variant.py"
t "trajopt 32" "X=1" "trajopt_variant.py --seeds 8"
t "generic trajopt 1024" "CUROBO_HIP_FUSED_NO_SHAPES=1" "trajopt_variant.py"
t "shapes coll" "X=1" "fused_variant.py --time"
cat $OUT/t.txt
python tests/randomised/fuzz_trajopt.py 120 77 2>&1 | tail -3
