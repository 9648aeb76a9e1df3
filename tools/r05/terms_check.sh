ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05_terms3
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_fused_shapes.py tests/test_gpu_trajopt.py tests/test_gpu_fused.py tests/test_gpu_parity_benchmarked.py tests/test_gpu_randomised_sweeps.py tests/test_gpu_sharded.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
t() { echo -n "$1 | " >> $OUT/t.txt; env $2 python $ROOT/tools/r05/$3 >> $OUT/t.txt 2>> $OUT/err.log; }
t "shapes coll" "X=1" "fused_variant.py --time"
t "shapes coll" "X=1" "fused_variant.py --time"
t "shapes coll 32" "X=1" "fused_variant.py --time --seeds 8"
t "generic coll" "CUROBO_HIP_FUSED_NO_SHAPES=1" "fused_variant.py --time"
t "shapes trajopt 1024" "X=1" "trajopt_variant.py"
t "shapes trajopt 32" "X=1" "trajopt_variant.py --seeds 8"
t "generic trajopt 1024" "CUROBO_HIP_FUSED_NO_SHAPES=1" "trajopt_variant.py"
cat $OUT/t.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --only c3,c5,fixed 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('other_configs_rollouts_per_s'))"
python -c "import json; d=json.load(open('bench_full.json')); print(d.get('full_trajopt_rollout')); print({k:v for k,v in d.get('trajopt_solve',{}).items() if k!='workload'})"
