#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call6; mkdir -p $O
python -m pytest tests/test_gpu_mesh.py -q -m gpu > $O/mesh_tests.log 2>&1; tail -5 $O/mesh_tests.log
python tools/r04/mesh_stats.py 2>&1 | tee $O/mesh_stats.txt
python -m pytest tests/test_gpu_parity_benchmarked.py -q -s -m gpu -k "per_sphere" > $O/parity_tests.log 2>&1
grep "c5 per sphere\|passed\|failed\|Error" $O/parity_tests.log | cut -c1-700
