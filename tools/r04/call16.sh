#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call16; mkdir -p $O
for ls in 1 2 4 8; do echo "leaf_size $ls"; CUROBO_MESH_LEAF_SIZE=$ls timeout 120 python tools/r04/mesh_stats.py 2>&1 | grep "per launch" ; done | tee $O/leaf.txt
