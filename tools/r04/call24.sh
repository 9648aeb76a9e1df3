#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call24; mkdir -p $O
cp curobo_amd/lib/variants/libcurobo_hip_rnea_pref.so curobo_amd/lib/libcurobo_hip.so
timeout 600 python -m pytest tests/test_gpu_dynamics.py -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
python tools/bench_rnea.py 2>&1 | tail -1
python tools/bench_rnea.py franka 33792 2>&1 | tail -1
python tools/probes/rnea_scratch_time.py 2>&1 | tail -2
