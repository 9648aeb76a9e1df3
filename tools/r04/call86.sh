#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for s in 21 22; do timeout 600 python tests/randomised/fuzz_fused.py 120 $s 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tail -5; done
