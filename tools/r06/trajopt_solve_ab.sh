# trajopt_solve of bench.py with the metrics pass captured (default) and eager, on the GPU box
cd $GRAFT_REPO_ROOT
for c in 1 0; do
CUROBO_CAPTURE_METRICS_PASS=$c python - <<PY 2>&1 | grep -v amdgpu.ids | tail -4
import json, torch, bench
from curobo_amd.kinematics import KinematicsCfg
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.workloads import c2_world
dev = torch.device("cuda:0")
kcfg = KinematicsCfg.from_packaged("franka", device=dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
r = bench.trajopt_solve_benchmark(kcfg.model, kcfg.kinematics_config, scene, dev, torch)
print("capture=$c", json.dumps({k: v for k, v in r.items() if k != "workload"}))
PY
done
