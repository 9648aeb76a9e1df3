#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call22; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_mppi.py tests/test_gpu_trajopt.py tests/test_gpu_parity_benchmarked.py -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
for nb in 0 1; do
  if [ $nb = 1 ]; then export CUROBO_HIP_NO_TABLES=1; fi
  echo "== NO_TABLES=$nb"
  python tools/profile_fused.py --batch 1024 2>&1 | grep "fused launch\|P0\|P1 FK\|P2 costs\|workgroup total"
  python bench.py --no-cpu-baseline --no-ik --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['readings']['exclusive_launch']['avg_launch_us'])"
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-ik --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver ', d['value'], d['ms_per_step'])"
done | tee $O/tables.txt
