#!/bin/bash
# round 4: the maximum-size tests (L-BFGS history, B-spline) first run
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call93; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_large_sizes.py -q -m gpu > $O/large.log 2>&1; tail -40 $O/large.log
