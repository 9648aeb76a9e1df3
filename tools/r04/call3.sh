#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call3
mkdir -p $O
python -m pytest tests/test_gpu_mesh.py -q -m gpu > $O/mesh_tests.log 2>&1; tail -5 $O/mesh_tests.log
python -m pytest tests/test_gpu_parity_benchmarked.py -q -s -m gpu -k "per_sphere" > $O/parity_tests.log 2>&1
grep "c5 per sphere\|passed\|failed\|Error" $O/parity_tests.log | cut -c1-600
python bench.py --only mesh --no-cpu-baseline > $O/bench_mesh.json 2> $O/bench_mesh.err; tail -3 $O/bench_mesh.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_call3/bench_mesh.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("mesh_world"), indent=1)[:2500])
print("ms_per_step", d.get("ms_per_step"))
PY
