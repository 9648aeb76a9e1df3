#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 800 python tests/randomised/fuzz_scene.py 80 1 2>&1 | grep -v amdgpu.ids | cut -c1-500 | tail -24
