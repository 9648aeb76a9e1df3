#!/bin/bash
cd "$GRAFT_REPO_ROOT"
cp curobo_amd/lib/variants/libcurobo_hip_meshstats.so curobo_amd/lib/libcurobo_hip.so
timeout 120 python tools/r04/mesh_stats.py 2>&1 | grep -v "per launch" | tail -8
