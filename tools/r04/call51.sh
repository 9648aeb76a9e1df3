#!/bin/bash
# mesh walk: a point inside the bounding box with nothing within the radius is searched within its distance to the nearest box face first
cd "$GRAFT_REPO_ROOT"
echo "== with the face bound"; timeout 180 python tools/r04/mesh_ab.py /tmp/fb.npz 2>&1 | grep "walk mode"
BATCH=256 timeout 180 python tools/r04/mesh_ab.py 2>&1 | grep "walk mode"
BATCH=4096 timeout 180 python tools/r04/mesh_ab.py 2>&1 | grep "walk mode"
cp curobo_amd/lib/libcurobo_hip.so /tmp/lib_default.so
cp curobo_amd/lib/variants/libcurobo_hip_nofb.so curobo_amd/lib/libcurobo_hip.so
echo "== without"; timeout 180 python tools/r04/mesh_ab.py /tmp/nofb.npz 2>&1 | grep "walk mode"
cp curobo_amd/lib/variants/libcurobo_hip_stats.so curobo_amd/lib/libcurobo_hip.so
echo "== counters (with the face bound)"; timeout 180 python tools/r04/mesh_stats.py 2>&1 | grep -v amdgpu.ids | tail -8
cp /tmp/lib_default.so curobo_amd/lib/libcurobo_hip.so
python tools/r04/mesh_ab.py --compare /tmp/fb.npz /tmp/nofb.npz
timeout 300 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -2
