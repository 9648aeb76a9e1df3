ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_seed_ik.py tests/test_gpu_ik.py tests/test_gpu_multiframe_ik.py tests/test_gpu_planner.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tests/randomised/fuzz_ik.py 30 11 2>&1 | grep -v amdgpu | tail -2
python bench.py --only ik --no-cpu-baseline --no-configs --steps 20 --warmup 5 > /dev/null 2>&1
python -c "import json; d=json.load(open('bench_full.json'))['ik']; print({k:d[k] for k in ('value','ms_per_batch','success_rate','solves_that_ran_lbfgs','median_position_error_m')}, d['full_optimizer']['ms_per_batch'])"
python bench.py --only ik --no-cpu-baseline --no-configs --steps 20 --warmup 5 > /dev/null 2>&1
python -c "import json; d=json.load(open('bench_full.json'))['ik']; print({k:d[k] for k in ('value','ms_per_batch','success_rate')})"
