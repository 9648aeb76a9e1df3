"""Exclusive fused-rollout launch on the C2 shapes with parts of the cost set switched off (timing / counters by elimination).

    python tools/r05/fused_variant.py [--no-self] [--no-scene] [--no-sweep] [--time] [--seeds N]

--time: HIP-event time of 20 launches replayed from a hipGraph (us per launch) printed as JSON; without it: ten plain launches
(the target of a rocprofv3 --pmc pass).  State = the line-search candidates of iteration 1 at the seeds, as bench.py's
exclusive-launch reading."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.workloads import c2_world, seed_knots, start_configuration  # noqa: E402

args = sys.argv[1:]
dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
cfg = CollisionRolloutCfg(use_self_collision="--no-self" not in args, use_scene_collision="--no-scene" not in args,
                          use_sweep="--no-sweep" not in args, use_speed_metric="--no-sweep" not in args)
seeds = int(args[args.index("--seeds") + 1]) if "--seeds" in args else 256
nls = 4
B = seeds * nls
ro = CollisionRollout(kin, scene, B, cfg)
ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
base = seed_knots(model, seeds, cfg.n_knots, seed=2)
rng = np.random.default_rng(0)
step = rng.normal(size=base.shape).astype(np.float32) * 0.02
knots = np.stack([base + a * step for a in (0.0, 0.1, 0.5, 1.0)], axis=1).reshape(B, -1)
x = torch.as_tensor(knots, device=dev)
for _ in range(10):
    c, g = ro.cost_and_gradient(x)
torch.cuda.synchronize()
if "--time" in args:
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
    print(json.dumps({"variant": " ".join(a for a in args if a.startswith("--no")) or "full", "B": B, "us_median": round(float(np.median(best)), 2),
                      "us_min": round(min(best), 2), "cost_sum": float(c.double().sum().item()), "grad_abs_sum": float(g.double().abs().sum().item())}))
