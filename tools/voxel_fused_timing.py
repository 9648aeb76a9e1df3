"""Fused vs kernel-sequence rollout on a mixed cuboid + ESDF voxel world (C5-like scene, C2 shapes):
the SWEEP x voxel instantiations of the fused kernel spill registers; is the fused launch still ahead?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curobo_amd.robot import load_packaged_robot  # noqa: E402
from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import CollisionRollout, CollisionRolloutCfg  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays, voxel_grid_from_sdf  # noqa: E402
from curobo_amd.workloads import c2_world, seed_knots, start_configuration  # noqa: E402

dev = torch.device("cuda:0")
model = load_packaged_robot("franka")
kin = KinematicsParams.from_model(model, dev)
arrays = cuboid_scene_arrays(c2_world())
sdf = lambda p: np.linalg.norm(p - np.array([0.35, -0.3, 0.5]), axis=-1) - 0.18  # noqa: E731
arrays = {**arrays, **voxel_grid_from_sdf(sdf, (64, 64, 64), 0.03, pose7=(0.1, -0.1, 0.5, 1, 0, 0, 0), max_distance=100.0)}
scene = SceneData.from_arrays(arrays, dev)
B = 1024
x = torch.as_tensor(seed_knots(model, B, 12, seed=2), device=dev).reshape(B, -1)
for fused in (True, False):
    ro = CollisionRollout(kin, scene, B, CollisionRolloutCfg(use_fused=fused))
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
    ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            ro.cost_and_gradient(x)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(("fused launch      " if fused else "kernel sequence   "), f"{e0.elapsed_time(e1) * 10:.1f} us per 1024 rollouts (cuboids + 64^3 ESDF, swept + speed metric)")
