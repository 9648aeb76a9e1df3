#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call2
mkdir -p $O
python -m pytest tests/test_gpu_parity_benchmarked.py -q -s -m gpu -k "per_sphere" > $O/parity_tests.log 2>&1
grep "c5 per sphere\|passed\|failed" $O/parity_tests.log
