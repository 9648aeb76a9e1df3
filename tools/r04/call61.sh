#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 200 python tools/ik_phases.py 2>&1 | grep -v amdgpu.ids | tail -12
