#!/bin/bash
cd "$GRAFT_REPO_ROOT"
X="--no-cpu-baseline --no-ik --no-configs"
for q in 8 16; do
for mode in "" "--joined-graphs"; do
  for cmdline in "--gpus 1 --steps 20 --warmup 5" ""; do
    GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py $cmdline $mode $X 2> /tmp/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('queues $q mode [$mode] cmd [$cmdline]:', d['value'], d['ms_per_step'], d['timing']['block_ms_median'], d['timing']['block_ms_min'], d['timing']['graphs'][:60])
" || tail -5 /tmp/err.log
  done
done
done
