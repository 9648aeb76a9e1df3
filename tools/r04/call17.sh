#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py -q -m gpu -x > $O/multirank.log 2>&1; tail -30 $O/multirank.log
CUROBO_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --selftest 2>&1 | tail -2 | tee $O/selftest.json
