#!/bin/bash
# HIP runtime knobs: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) -- headline, driver command, solver latencies
cd "$GRAFT_REPO_ROOT"
for k in 0 1; do
  for cmdline in "--gpus 1 --steps 20 --warmup 5" ""; do
    HIP_FORCE_DEV_KERNARG=$k timeout 250 python bench.py $cmdline --no-cpu-baseline --only ik 2> /tmp/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('DEV_KERNARG=$k cmd [$cmdline]:', d['value'], d['ms_per_step'], d['timing']['block_ms_median'], 'ik', d.get('ik',{}).get('ms_per_batch'), d.get('ik',{}).get('full_optimizer',{}).get('ms_per_batch'), 'trajopt', {k2:v.get('ms_per_batch') for k2,v in d.get('trajopt_solve',{}).items() if isinstance(v,dict)}, 'fullrollout', d.get('full_trajopt_rollout',{}).get('fused_us'))
" || tail -5 /tmp/err.log
  done
done
