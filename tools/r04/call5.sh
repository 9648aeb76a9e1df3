#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call5; mkdir -p $O
cp curobo_amd/lib/libcurobo_hip.so /tmp/orig.so
cp curobo_amd/lib/variants/libcurobo_hip_meshstats.so curobo_amd/lib/libcurobo_hip.so
python tools/r04/mesh_stats.py 2>&1 | tee $O/mesh_stats.txt
cp /tmp/orig.so curobo_amd/lib/libcurobo_hip.so
python tools/r04/mesh_stats.py 2>&1 | tee -a $O/mesh_stats.txt
