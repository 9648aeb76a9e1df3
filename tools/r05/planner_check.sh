ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_planner.py tests/test_gpu_trajopt.py tests/test_gpu_multiframe_ik.py tests/test_gpu_ik.py tests/test_gpu_api.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -4
python tests/randomised/fuzz_planner.py 6 31 2>&1 | tail -4
python bench.py --only fixed,ik --no-cpu-baseline --no-configs --steps 20 --warmup 5 > /dev/null 2>&1
python -c "import json; d=json.load(open('bench_full.json')); print({k:(v.get('ms_per_batch'), v.get('success_rate'), v.get('with_convergence_exit',{}).get('ms_per_batch')) for k,v in d.get('trajopt_solve',{}).items() if isinstance(v,dict)})"
