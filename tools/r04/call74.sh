#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 800 python tests/randomised/fuzz_fk_bspline.py 60 1 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tail -24
