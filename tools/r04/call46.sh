#!/bin/bash
# mesh queue with the heavy spheres (centre inside a live mesh's bounding box) first
cd "$GRAFT_REPO_ROOT"
echo "== heavy first"; timeout 180 python tools/r04/mesh_ab.py /tmp/hf.npz 2>&1 | grep "walk mode"
BATCH=256 timeout 180 python tools/r04/mesh_ab.py 2>&1 | grep "walk mode"
cp curobo_amd/lib/libcurobo_hip.so /tmp/lib_default.so
cp curobo_amd/lib/variants/libcurobo_hip_nohf.so curobo_amd/lib/libcurobo_hip.so
echo "== one class (reverse order)"; timeout 180 python tools/r04/mesh_ab.py /tmp/nohf.npz 2>&1 | grep "walk mode"
cp /tmp/lib_default.so curobo_amd/lib/libcurobo_hip.so
python tools/r04/mesh_ab.py --compare /tmp/hf.npz /tmp/nohf.npz
timeout 300 python -m pytest tests/test_gpu_mesh.py -q -m gpu -x 2>&1 | tail -2
