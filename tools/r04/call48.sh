#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 400 python -X faulthandler tools/r04/c4_shards.py 1 2 4 8 2>&1 | grep -v amdgpu.ids | tail -30
