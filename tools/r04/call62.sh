#!/bin/bash
# the fixed cost of a timed block: one graph of K iterations (K = 5 .. 80), and lead-ins of 1 .. 10 iterations at K = 20
cd "$GRAFT_REPO_ROOT"
X="--no-cpu-baseline --no-ik --no-configs --gpus 1 --warmup 5"
run() { timeout 200 python bench.py $X "$@" 2> /tmp/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '->', d['ms_per_step'], 'block', d['timing']['block_ms_median'], 'min', d['timing']['block_ms_min'])
" || tail -3 /tmp/err.log; }
for k in 5 10 20 40 80; do run --steps $k --graph-lead 0 --graph-iters $k; done
for l in 1 2 3 5 8 10; do run --steps 20 --graph-lead $l --graph-iters 25; done
