#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for seed in 2 3 4; do timeout 500 python tests/randomised/fuzz_fused.py 150 $seed 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tail -6; done
