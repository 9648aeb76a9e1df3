#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python tests/randomised/fuzz_trajopt.py 60 1 2>&1 | grep -v amdgpu.ids | cut -c1-900 | tail -30
