#!/bin/bash
cd "$GRAFT_REPO_ROOT"
cp curobo_amd/lib/variants/libcurobo_hip_stats.so curobo_amd/lib/libcurobo_hip.so
timeout 300 python tools/r04/mesh_heavy.py 2>&1 | grep -v amdgpu.ids | tail -5
