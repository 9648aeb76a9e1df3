#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python tests/randomised/fuzz_mesh.py 24 1 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tail -24
