#!/bin/bash
# round 4: mesh launch against the BVH leaf size (development knob CUROBO_MESH_LEAF_SIZE; 8 is the default)
cd "$GRAFT_REPO_ROOT"
for ls in 8 6 12 16 24; do
  echo "== leaf size $ls"; CUROBO_MESH_LEAF_SIZE=$ls timeout 60 python tools/r04/mesh_ab.py /tmp/l$ls.npz 2>&1 | grep "walk mode"
done
for ls in 6 12 16 24; do python tools/r04/mesh_ab.py --compare /tmp/l8.npz /tmp/l$ls.npz | tr '\n' ';'; echo; done
