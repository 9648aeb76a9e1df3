#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 500 python tests/randomised/fuzz_fused.py 80 1 2>&1 | grep -v amdgpu.ids | tail -40
