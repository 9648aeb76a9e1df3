#!/bin/bash
# round 4, checkpoint D (session 3): full GPU suite, both bench lines, kernel stats + counters of the mesh kernels (the only kernels changed since r04_c)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call54; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python - <<'PY'
import json
for f in ("bench_default","bench_driver"):
    try:
        d=json.loads(open(f"gpurun_out/r04_call54/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("mesh_world",{}).get("mesh_launch",{}).get("us"), d.get("mesh_world",{}).get("mesh_launch",{}).get("slowdown_vs_cuboid_kernel"), d.get("c4_humanoid_share",{}).get("us_per_rollout_set"))
    except Exception as e: print(f, "ERR", e)
PY
COUNTERS_ONLY=1 timeout 900 bash tools/collect_profiles_r04.sh r04_d mesh > $O/collect.log 2>&1; tail -5 $O/collect.log
