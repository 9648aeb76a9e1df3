#!/bin/bash
# bench.py with 8 gloo ranks sharing the one GPU: the multi-rank logic at the driver's largest world size (functional check only)
cd "$GRAFT_REPO_ROOT"
export CUROBO_BENCH_BACKEND=gloo
( time timeout 400 python bench.py --gpus 8 --selftest ) 2>&1 | grep -v amdgpu.ids | tail -4
( time timeout 600 python bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline ) > gpurun_out/bench_gloo8.log 2>&1
python - <<'PY'
import json
t=open("gpurun_out/bench_gloo8.log").read()
line=[l for l in t.splitlines() if l.startswith("{")]
print(t[-600:] if not line else "")
if line:
    d=json.loads(line[-1])
    print("n_gpus", d["n_gpus"], "value", d["value"], "ms_per_step", d["ms_per_step"], "scaling", d["scaling"], "strong", {k:d.get("strong_scaling",{}).get(k) for k in ("value","ms_per_step","error")}, "legs", {k:(v.get("value") or v.get("error")) for k,v in d.get("multi_gpu_legs",{}).items()})
print([l for l in t.splitlines() if l.startswith("real")])
PY
