#!/bin/bash
# bench: one linear graph per seed shard (no join between the graphs) against one graph over all shards
cd "$GRAFT_REPO_ROOT"
X="--no-cpu-baseline --no-ik --no-configs"
for mode in "" "--joined-graphs"; do
  for cmdline in "--gpus 1 --steps 20 --warmup 5" ""; do
    timeout 200 python bench.py $cmdline $mode $X 2> /tmp/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mode [$mode] cmd [$cmdline]:', d['value'], d['ms_per_step'], d['timing']['block_ms_median'], d['timing']['block_ms_min'], d['best_cost'], d['best_seed'], d['timing']['graphs'][:80])
" || tail -5 /tmp/err.log
  done
done
