#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call33; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mesh.py -q -m gpu -x > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 120 python tools/r04/mesh_stats.py 2>&1 | grep "per launch"
