#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_randomised_sweeps.py -q -m gpu 2>&1 | tail -8
