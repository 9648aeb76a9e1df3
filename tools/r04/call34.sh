#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in mb16384 mb32768 mb65536; do
cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so
echo "$v"; timeout 120 python tools/r04/mesh_stats.py 2>&1 | grep "per launch" 
done
