#!/bin/bash
cd "$GRAFT_REPO_ROOT"
X="--no-cpu-baseline --no-ik --no-configs"
run() { timeout 200 python bench.py $X "$@" 2> /tmp/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '->', d['ms_per_step'], 'block', d['timing']['block_ms_median'], 'min', d['timing']['block_ms_min'])
" || tail -3 /tmp/err.log; }
for s in 4 2 8; do run --gpus 1 --steps 20 --warmup 5 --shards $s; run --shards $s; done
run --gpus 1 --steps 20 --warmup 5 --graph-lead 3
run --gpus 1 --steps 20 --warmup 5 --graph-lead 8
