#!/bin/bash
# check of the final code (select kernel rewrite, large-size tests): GPU suite + both bench lines
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call99; mkdir -p $O
( time timeout 900 python -m pytest tests -q -m gpu ) > $O/gpu_suite.log 2>&1; tail -6 $O/gpu_suite.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python - <<'PY'
import json
for f in ("bench_default","bench_driver"):
    try:
        d=json.loads(open(f"gpurun_out/r04_call99/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("mesh_world",{}).get("mesh_launch",{}).get("us"), d.get("c4_humanoid_share",{}).get("us_per_rollout_set"), d.get("ik",{}).get("value"), d.get("trajopt_solve",{}).get("1_problems_x_8_seeds",{}).get("ms_per_batch"))
    except Exception as e: print(f, "ERR", e)
PY
