ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_fused_shapes.py tests/test_gpu_fused.py tests/test_gpu_trajopt.py tests/test_gpu_randomised_sweeps.py -m gpu -q 2>&1 | tail -3
CUROBO_HIP_JIT_SHAPES=1 timeout 1500 python tests/randomised/fuzz_fused.py 60 509 2>&1 | grep -v amdgpu | tail -3
timeout 600 python tests/randomised/fuzz_fk_bspline.py 200 605 2>&1 | grep -v amdgpu | tail -2
timeout 600 python tests/randomised/fuzz_fused.py 200 606 2>&1 | grep -v amdgpu | tail -1
python tools/r05/fused_variant.py --time 2>&1 | grep -v amdgpu
