#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call12; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_mesh.py -q -m gpu -x > $O/mesh_tests.log 2>&1; tail -5 $O/mesh_tests.log
timeout 120 python tools/r04/mesh_stats.py 2>&1 | grep -v amdgpu.ids | tee $O/mesh_stats.txt
cp curobo_amd/lib/variants/libcurobo_hip_meshstats.so curobo_amd/lib/libcurobo_hip.so
timeout 120 python tools/r04/mesh_stats.py 2>&1 | grep "closest\|per lane" | tee -a $O/mesh_stats.txt
