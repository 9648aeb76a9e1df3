"""Phase stamps of the fused launch with the full trajectory-optimisation cost set (needs a library whose TERMS kernels were built
with -DCUROBO_FUSED_STAMP_TERMS).  Slots as tools/r05/phase_stamps.py, plus 12 -> 13 tool pose, 13 -> 14 c-space STATE, 14 -> 3 gather."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd._lib import load  # noqa: E402
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.workloads import c2_world, seed_knots, start_configuration  # noqa: E402

args = sys.argv[1:]
seeds = int(args[args.index("--seeds") + 1]) if "--seeds" in args else 8
K = 10
dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
B = seeds * 4
ro = TrajOptRollout(kin, scene, B, TrajOptRolloutCfg())
ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
x = torch.as_tensor(seed_knots(model, B, 12, seed=2), device=dev).reshape(B, -1)
for _ in range(10):
    ro.cost_and_gradient(x)
torch.cuda.synchronize()
lib = load()
buf = torch.zeros((K, B, 16), dtype=torch.int64, device=dev)
lib.curobo_hip_rollout_fused_set_profile_sequence(buf.data_ptr(), K, B)
for _ in range(K):
    ro.cost_and_gradient(x)
torch.cuda.synchronize()
lib.curobo_hip_rollout_fused_set_profile_sequence(None, 0, 0)
t = buf[2:].cpu().numpy().astype(np.float64) / 100.0
order = [0, 1, 8, 9, 10, 2, 7, 12, 13, 14, 3, 4]
names = {1: "tables+bspline", 8: "sincos", 9: "fk_chains", 10: "spheres", 2: "barrier", 7: "main_round", 12: "leftover+wait", 13: "tool_pose",
         14: "cspace_state", 3: "gather", 4: "bspline_vjp+out"}
out = {"B": B}
prev = 0
for s in order[1:]:
    ok = (t[:, :, s] > 0) & (t[:, :, prev] > 0)
    out[names[s]] = round(float((t[:, :, s] - t[:, :, prev])[ok].mean()), 2) if ok.any() else None
    if ok.any():
        prev = s
out["workgroup_total"] = round(float((t[:, :, 4] - t[:, :, 0]).mean()), 2)
print(json.dumps(out))
