#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call32; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mesh.py -q -m gpu -x -s -k "mesh_world or kernel_sequence" > $O/tests.log 2>&1; tail -30 $O/tests.log
