#!/bin/bash
# round 4, checkpoint C: full GPU suite, default + driver bench lines, profile collection (kernel stats + counters)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call40; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python - <<'PY'
import json
for f in ("bench_default","bench_driver"):
    try:
        d=json.loads(open(f"gpurun_out/r04_call40/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v.get("us_per_rollout_set") or v.get("fused",{}).get("us_per_launch_set")) for k,v in d.items() if k.startswith("c") and isinstance(v,dict) and ("us_per_rollout_set" in v or "fused" in v)}, d.get("mesh_world",{}).get("mesh_launch",{}).get("us"), d.get("trajopt_solve"))
    except Exception as e: print(f, "ERR", e)
PY
timeout 1500 bash tools/collect_profiles_r04.sh r04_c > $O/collect.log 2>&1; tail -5 $O/collect.log
