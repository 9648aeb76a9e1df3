#!/bin/bash
# round 4: select kernel at 4 spheres per lane (the default now): mesh tests (incl. the per-sphere slot reads), mesh fuzzer, launch time
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call96; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_mesh.py -q -m gpu > $O/mesh_tests.log 2>&1; tail -5 $O/mesh_tests.log
timeout 120 python tests/randomised/fuzz_mesh.py 12 31 > $O/fuzz_mesh.log 2>&1; tail -2 $O/fuzz_mesh.log
timeout 120 python tools/r04/mesh_ab.py 2>&1 | grep "walk mode"
