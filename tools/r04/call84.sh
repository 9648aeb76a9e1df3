#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python tests/randomised/fuzz_mppi.py 80 1 2>&1 | grep -v amdgpu.ids | cut -c1-500 | tail -20
