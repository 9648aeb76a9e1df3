#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_fused.py -q -m gpu -x -k "randomly_rotated" 2>&1 | tail -8
