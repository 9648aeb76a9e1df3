#!/bin/bash
# Morton keys of the mesh build: one scale for the three axes against a scale per axis
cd "$GRAFT_REPO_ROOT"
echo "== one scale"; timeout 180 python tools/r04/mesh_ab.py /tmp/a.npz 2>&1 | grep "walk mode"
BATCH=256 timeout 180 python tools/r04/mesh_ab.py 2>&1 | grep "walk mode"
echo "== per axis"; CUROBO_MESH_MORTON_PER_AXIS=1 timeout 180 python tools/r04/mesh_ab.py /tmp/b.npz 2>&1 | grep "walk mode"
python tools/r04/mesh_ab.py --compare /tmp/a.npz /tmp/b.npz
timeout 300 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -2
