#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_mesh.py -q -m gpu -x -k "queued" 2>&1 | tail -15
