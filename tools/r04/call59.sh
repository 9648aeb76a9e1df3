#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 200 python tools/r04/trajopt_phases.py 2>&1 | grep -v amdgpu.ids | tail -12
P=64 S=4 timeout 200 python tools/r04/trajopt_phases.py 2>&1 | grep -v amdgpu.ids | tail -12
