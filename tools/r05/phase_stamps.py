"""Phase stamps of the fused C2 launch (the library's profile sequence: every workgroup's thread 0 writes the device wall clock,
100 MHz, at the phase boundaries) at a given batch: where the time of ONE trajectory goes when the chip is empty (small
batches = a latency chain) and when it is full.

    python tools/r05/phase_stamps.py [--seeds N] [--launches K]

Slots: 0 start, 1 tables + B-spline staged, 8 joint sin/cos, 9 FK chains, 10 spheres, 2 = P2 begins, 5 / 6 = self-collision done /
scene pass done of point b % H (its wavefront), 7 main round done, 12 link wrenches folded, 3 FK VJP done, 4 end."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd._lib import load  # noqa: E402
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.workloads import c2_world, seed_knots, start_configuration  # noqa: E402

args = sys.argv[1:]
seeds = int(args[args.index("--seeds") + 1]) if "--seeds" in args else 8
K = int(args[args.index("--launches") + 1]) if "--launches" in args else 10
dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
B = seeds * 4
ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg(longest_first_dispatch="--no-reorder" not in args))
ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
base = seed_knots(model, seeds, ro.cfg.n_knots, seed=2)
rng = np.random.default_rng(0)
step = rng.normal(size=base.shape).astype(np.float32) * 0.02
x = torch.as_tensor(np.stack([base + a * step for a in (0.0, 0.1, 0.5, 1.0)], axis=1).reshape(B, -1), device=dev)
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
t = buf[2:].cpu().numpy().astype(np.float64) / 100.0  # us; [launch, workgroup, slot]
order = [0, 1, 8, 9, 10, 2, 7, 12, 3, 4]
names = {1: "tables+bspline", 8: "sincos", 9: "fk_chains", 10: "spheres", 2: "barrier", 7: "main_round(self+scene)", 12: "wrench_fold", 3: "fk_vjp", 4: "bspline_vjp+out"}
out = {"B": B, "reorder": "--no-reorder" not in args, "threads": int(os.environ.get("CUROBO_HIP_FUSED_THREADS", 0)) or "default"}
prev = 0
for s in order[1:]:
    ok = (t[:, :, s] > 0) & (t[:, :, prev] > 0)
    out[names[s]] = round(float((t[:, :, s] - t[:, :, prev])[ok].mean()), 2) if ok.any() else None
    if ok.any():
        prev = s
ok = (t[:, :, 5] > 0) & (t[:, :, 6] > 0)
out["one_point_self"] = round(float((t[:, :, 5] - t[:, :, 2])[ok].mean()), 2)
out["one_point_scene"] = round(float((t[:, :, 6] - t[:, :, 5])[ok].mean()), 2)
out["workgroup_total"] = round(float((t[:, :, 4] - t[:, :, 0]).mean()), 2)
out["launch_span"] = round(float((t[:, :, 4].max(axis=1) - t[:, :, 0].min(axis=1)).mean()), 2)
print(json.dumps(out))
