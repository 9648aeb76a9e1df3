"""Exclusive launch time of the FULL trajopt cost set (pose + c-space STATE + self + swept scene) in the fused launch, C2 shapes.
    python tools/r05/trajopt_variant.py [--seeds N]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.workloads import c2_world, seed_knots, start_configuration  # noqa: E402

args = sys.argv[1:]
seeds = int(args[args.index("--seeds") + 1]) if "--seeds" in args else 256
dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
B = seeds * 4
ro = TrajOptRollout(kin, scene, B, TrajOptRolloutCfg())
ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
x = torch.as_tensor(seed_knots(model, B, 12, seed=2), device=dev).reshape(B, -1)
for _ in range(10):
    c, g = ro.cost_and_gradient(x)
torch.cuda.synchronize()
reps = 20
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(reps):
        ro.cost_and_gradient(x)
for _ in range(3):
    gr.replay()
best = []
for _ in range(15):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) * 1e3 / reps)
print(json.dumps({"B": B, "us_median": round(float(np.median(best)), 2), "us_min": round(min(best), 2),
                  "cost_sum": float(c.double().sum().item()), "grad_abs_sum": float(g.double().abs().sum().item())}))
