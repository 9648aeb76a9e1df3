#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python tests/randomised/fuzz_opt.py 80 1 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tail -24
for s in 2 3; do timeout 600 python tests/randomised/fuzz_trajopt.py 80 $s 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tail -6; done
