#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for lead in 0 2 3 4 5 6 8; do
  for rep in 1 2; do
  timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --graph-lead $lead --no-cpu-baseline --no-ik --no-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lead', $lead, d['ms_per_step'], d['value'], d['timing']['block_ms_median'])"
  done
done
