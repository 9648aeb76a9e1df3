#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call26; mkdir -p $O
for v in g8 g16; do
cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so
timeout 900 python -m pytest tests/test_gpu_mesh.py -q -m gpu -x > $O/tests_$v.log 2>&1; tail -1 $O/tests_$v.log
for ls in 8 16 32; do echo "$v leaf_size $ls"; CUROBO_MESH_LEAF_SIZE=$ls timeout 120 python tools/r04/mesh_stats.py 2>&1 | grep "per launch" ; done | tee -a $O/leaf.txt
done
