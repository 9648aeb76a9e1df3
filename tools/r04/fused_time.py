"""Exclusive fused-rollout launch on the C2 workload (1024 trajectories at the seed state): median launch time over HIP events
and a checksum of its outputs (to compare library variants bit for bit).   python tools/r04/fused_time.py [batch]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.workloads import c2_world, seed_knots, start_configuration  # noqa: E402

dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
cfg = CollisionRolloutCfg(use_fused=True)
for B in [int(a) for a in sys.argv[1:]] or [1024]:
    ro = CollisionRollout(kin, scene, B, cfg)
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
    x = torch.as_tensor(seed_knots(model, B, cfg.n_knots, seed=2), device=dev).reshape(B, -1)
    for _ in range(20):
        out = ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    n = 300
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        out = ro.cost_and_gradient(x)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
    h = hashlib.sha256()
    for t in out:
        h.update(t.detach().cpu().numpy().tobytes())
    print(f"batch {B}: median {ts[n // 2]:.2f} us, p10 {ts[n // 10]:.2f} us, min {ts[0]:.2f} us per launch; outputs sha256 {h.hexdigest()[:16]}", flush=True)
