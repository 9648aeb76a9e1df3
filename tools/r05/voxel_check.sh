ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for i in 1 2; do
python bench.py --only c3,c5 --no-cpu-baseline --no-ik --steps 20 --warmup 5 > /dev/null 2>&1
python -c "import json; f=json.load(open('bench_full.json')); print('c3 fused', f['c3_ur10e_voxel']['fused']['us_per_launch_set'], 'seq scene', f['c3_ur10e_voxel']['kernels']['scene_collision_voxel_swept']['us'], 'c5 mixed', f['c5_batch_planner_share']['mixed cuboid + ESDF']['fused']['us_per_launch_set'], 'cuboid', f['c5_batch_planner_share']['cuboid-only']['fused']['us_per_launch_set'])"
done
timeout 900 python -m pytest tests/test_gpu_parity_benchmarked.py tests/test_gpu_fused.py tests/test_fused_shapes.py tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -2
