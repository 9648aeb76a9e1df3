#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/profile_fused.py 2>&1 | grep -v amdgpu.ids | tail -40
