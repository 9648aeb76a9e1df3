"""Torque-limited trajopt rollout (Franka, C2 world): one fused launch vs the kernel sequence with the joint-space chain on a side
stream, us per cost + gradient at several batch sizes."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch, numpy as np
import bench as B_
from conftest import load_model
from curobo_amd.robot.kinematics_params import KinematicsParams
from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
from curobo_amd.scene import SceneData, cuboid_scene_arrays
from curobo_amd.workloads import c2_world, seed_knots, start_configuration
dev = torch.device("cuda:0")
model = load_model("franka"); kin = KinematicsParams.from_model(model, dev)
scene = SceneData.from_arrays(cuboid_scene_arrays(c2_world()), dev)
for B, fused in [(b, f) for b in (32, 128, 512, 1024, 4096) for f in (True, False)]:
    cfg = TrajOptRolloutCfg(use_fused=fused, use_torque_limits=True)
    ro = TrajOptRollout(kin, scene, B, cfg)
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
    knots = torch.as_tensor(seed_knots(model, B, cfg.n_knots, seed=1, spread=0.5), device=dev).reshape(B, -1)
    fn = lambda: ro.cost_and_gradient(knots)
    c, g = fn(); torch.cuda.synchronize()
    gr = B_.graphed(fn, 3, torch)
    t = B_.time_kernel(gr.replay, 3, torch, min_s=0.05) / 3
    print(B, "fused" if fused else "sequence", round(t, 1), "us", float(c.sum()), float(g.abs().sum()))
