#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call23; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dynamics.py tests/test_gpu_trajopt.py tests/test_gpu_mppi.py -q -s -m gpu > $O/tests.log 2>&1; grep "c4 parity\|g1 rollout" $O/tests.log | cut -c1-260; tail -3 $O/tests.log
cp curobo_amd/lib/libcurobo_hip.so /tmp/new.so
for v in before new; do
  if [ $v = before ]; then cp curobo_amd/lib/variants/libcurobo_hip_before_rnea.so curobo_amd/lib/libcurobo_hip.so; else cp /tmp/new.so curobo_amd/lib/libcurobo_hip.so; fi
  echo "== $v"
  python tools/bench_rnea.py 2>&1 | tail -1
  python tools/bench_rnea.py franka 33792 2>&1 | tail -1
  python tools/probes/rnea_scratch_time.py 2>&1 | tail -4
  python bench.py --only c4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['c4_humanoid_share']
print('c4 set us', d['us_per_rollout_set'], {k:v['us'] for k,v in d['kernels'].items()})"
done | tee $O/rnea.txt
