#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call28; mkdir -p $O
for v in g8 w4 w5 w6; do
cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so
echo "$v"; timeout 120 python tools/r04/mesh_stats.py 2>&1 | grep "per launch" 
done | tee $O/occ.txt
