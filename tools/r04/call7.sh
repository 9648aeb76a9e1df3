#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call7; mkdir -p $O
python -m pytest tests/test_gpu_mesh.py -q -m gpu > $O/mesh_tests.log 2>&1; tail -3 $O/mesh_tests.log
python tools/r04/mesh_stats.py 2>&1 | tee $O/mesh_stats.txt
cp curobo_amd/lib/libcurobo_hip.so /tmp/orig.so
cp curobo_amd/lib/variants/libcurobo_hip_meshstats.so curobo_amd/lib/libcurobo_hip.so
python tools/r04/mesh_stats.py 2>&1 | grep "closest" | tee -a $O/mesh_stats.txt
cp /tmp/orig.so curobo_amd/lib/libcurobo_hip.so
python -m pytest tests/test_gpu_parity_benchmarked.py -q -s -m gpu -k "per_sphere and rotated" > $O/parity_tests.log 2>&1
grep "fused launch vs\|passed\|failed\|Error" $O/parity_tests.log | cut -c1-400
