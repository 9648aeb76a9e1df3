#!/bin/bash
# HSA completion signals by polling instead of interrupts (HSA_ENABLE_INTERRUPT=0): the timed block's final synchronize
cd "$GRAFT_REPO_ROOT"
X="--no-cpu-baseline --no-ik --no-configs"
for k in 1 0 1 0; do
  for cmdline in "--gpus 1 --steps 20 --warmup 5" ""; do
    HSA_ENABLE_INTERRUPT=$k timeout 250 python bench.py $cmdline $X 2> /tmp/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('HSA_ENABLE_INTERRUPT=$k cmd [$cmdline]:', d['value'], d['ms_per_step'], d['timing']['block_ms_median'], d['timing']['block_ms_min'])
" || tail -5 /tmp/err.log
  done
done
