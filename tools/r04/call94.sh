#!/bin/bash
# round 4: select kernel with the obstacle slots staged in LDS -- A/B against the per-sphere loads, outputs bitwise, mesh tests
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call94; mkdir -p $O
cp curobo_amd/lib/libcurobo_hip.so /tmp/lib_default.so
echo "== staged"; timeout 120 python tools/r04/mesh_ab.py /tmp/a.npz 2>&1 | grep "walk mode"
cp curobo_amd/lib/variants/libcurobo_hip_unstaged.so curobo_amd/lib/libcurobo_hip.so
echo "== unstaged"; timeout 120 python tools/r04/mesh_ab.py /tmp/b.npz 2>&1 | grep "walk mode"
cp /tmp/lib_default.so curobo_amd/lib/libcurobo_hip.so
python tools/r04/mesh_ab.py --compare /tmp/a.npz /tmp/b.npz
timeout 300 python -m pytest tests/test_gpu_mesh.py -q -m gpu -x > $O/mesh_tests.log 2>&1; tail -3 $O/mesh_tests.log
cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof -o m -- python $GRAFT_REPO_ROOT/tools/r04/mesh_ab.py > /dev/null 2>&1; grep -h "mesh" /tmp/prof/*/*kernel_stats.csv /tmp/prof/*kernel_stats.csv 2>/dev/null | cut -c1-200 | head -4
