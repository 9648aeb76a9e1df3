#!/bin/bash
# round 4, session 3: the self-scheduled mesh walk against the eight-at-a-time walk (bitwise + timing), work counters, mesh tests
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_call41; mkdir -p $O
export CUROBO_MESH_WALK=1; timeout 180 python tools/r04/mesh_ab.py $O/walk1.npz > $O/walk1.log 2>&1; tail -3 $O/walk1.log
export CUROBO_MESH_WALK=2; timeout 180 python tools/r04/mesh_ab.py $O/walk2.npz > $O/walk2.log 2>&1; tail -3 $O/walk2.log
python tools/r04/mesh_ab.py --compare $O/walk1.npz $O/walk2.npz
cp curobo_amd/lib/libcurobo_hip.so /tmp/lib_default.so
for v in claim4 stats; do
  cp curobo_amd/lib/variants/libcurobo_hip_$v.so curobo_amd/lib/libcurobo_hip.so
  for m in 2 1; do
    [ $v = claim4 ] && [ $m = 1 ] && continue
    CUROBO_MESH_WALK=$m timeout 180 python tools/r04/mesh_ab.py > $O/${v}_$m.log 2>&1; echo "== $v mode $m"; tail -5 $O/${v}_$m.log
  done
done
cp /tmp/lib_default.so curobo_amd/lib/libcurobo_hip.so
unset CUROBO_MESH_WALK
timeout 300 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_kernels.py -q -m gpu -x > $O/mesh_tests.log 2>&1; tail -3 $O/mesh_tests.log
timeout 400 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
timeout 200 python bench.py --only mesh > $O/bench_mesh.json 2> $O/bench_mesh.err; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r04_call41/bench_mesh.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], json.dumps(d.get("mesh_world"))[:600])
except Exception as e: print("ERR", e)
PY
