#!/bin/bash
cd "$GRAFT_REPO_ROOT"
cp curobo_amd/lib/libcurobo_hip.so /tmp/keep.so
for v in w5 w6 w8; do
cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so
echo "$v"; timeout 120 python tools/r04/mesh_stats.py 2>&1 | grep "per launch" 
done
