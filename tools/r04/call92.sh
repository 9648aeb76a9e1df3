#!/bin/bash
# round 4: the maximum-size test alone (first run), then a few fuzzers on fresh seeds
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call92; mkdir -p $O
timeout 240 python -m pytest tests/test_gpu_large_sizes.py -q -m gpu -x > $O/large.log 2>&1; tail -25 $O/large.log
timeout 60 python tests/randomised/fuzz_scene.py 30 77 > $O/fuzz_scene.log 2>&1; tail -2 $O/fuzz_scene.log
timeout 60 python tests/randomised/fuzz_fused.py 12 77 > $O/fuzz_fused.log 2>&1; tail -2 $O/fuzz_fused.log
