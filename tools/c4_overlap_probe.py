"""C4 rollout set (Unitree G1, kernel sequence): task-space chain alone, joint-space chain on the same stream, on a side stream (development probe)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B_
from curobo_amd.kinematics import KinematicsCfg
from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg
from curobo_amd.workloads import seed_knots, start_configuration
dev = torch.device("cuda:0")
kcfg = KinematicsCfg.from_packaged("unitree_g1", device=dev)
model, kin = kcfg.model, kcfg.kinematics_config
B = 1024
x = torch.as_tensor(seed_knots(model, B, 12, seed=6, spread=0.15), device=dev).reshape(B, -1)
for name, kw in (("task chain only (no torque limits)", dict(use_torque_limits=False)),
                 ("one stream", dict(use_torque_limits=True, effort_limit=[200.0] * kin.num_dof, overlap_dynamics=False)),
                 ("side stream", dict(use_torque_limits=True, effort_limit=[200.0] * kin.num_dof, overlap_dynamics=True))):
    ro = TrajOptRollout(kin, None, B, TrajOptRolloutCfg(use_fused=False, **kw))
    ro.update_start_state(torch.as_tensor(start_configuration(model), device=dev))
    ro.cost_and_gradient(x); ro.cost_and_gradient(x)
    torch.cuda.synchronize()
    g = B_.graphed(lambda: ro.cost_and_gradient(x), 2, torch)
    us = B_.time_kernel(g.replay, 2, torch, min_s=0.1) / 2
    print(f"{name:40s} {us:8.1f} us")
