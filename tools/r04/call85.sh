#!/bin/bash
# the GPU suite with every randomised sweep in it
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call85; mkdir -p $O
( time timeout 900 python -m pytest tests -q -m gpu ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log
