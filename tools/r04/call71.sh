#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 800 python tests/randomised/fuzz_self.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tail -14
