#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python -X faulthandler -m pytest tests/test_gpu_trajopt.py -q -m gpu -x -k "seed_shards_over_torque" 2>&1 | tail -15
timeout 300 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_ik.py -q -m gpu -x 2>&1 | tail -3
