"""full trajectory-optimisation cost set on the dual UR10e (12 dof, two tool frames): fused launch against the kernel sequence on the
same knots (cost, gradient, per-term outputs) with the swept scene term on and off, collision_table world"""
import sys
import numpy as np
import torch

sys.path.insert(0, "tests")
from conftest import load_model, sample_q  # noqa: E402

from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg  # noqa: E402
from curobo_amd.scene.config import scene_from_config  # noqa: E402
from curobo_amd.workloads import seed_knots  # noqa: E402

dev = "cuda:0"
for robot in sys.argv[1:] or ("dual_ur10e",):
    model = load_model(robot)
    kin = KinematicsParams.from_model(model, dev)
    scene = scene_from_config("collision_table.yml", dev)
    D, T = kin.num_dof, kin.num_pose_links
    for sweep in (False, True):
        cfgs = [TrajOptRolloutCfg(use_sweep=sweep, use_speed_metric=sweep, use_fused=f) for f in (True, False)]
        B, nk = 12, cfgs[0].n_knots
        knots = torch.as_tensor(seed_knots(model, B, nk, seed=5, spread=0.5), device=dev).reshape(B, -1)
        q = sample_q(model, 3, seed=8, scale=0.5)
        out = []
        for cfg in cfgs:
            ro = TrajOptRollout(kin, scene, B, cfg)
            start = torch.as_tensor(q[0], device=dev)
            ro.update_start_state(start)
            from curobo_amd.kinematics import Kinematics, KinematicsCfg
            if "goal" not in globals():
                kk = Kinematics(KinematicsCfg.from_packaged(robot, dev))
                tp = kk.compute_kinematics(torch.as_tensor(q[1:3], device=dev)).tool_poses
                goal = (tp.position[:, 0].reshape(2, T, 1, 3).clone(), tp.quaternion[:, 0].reshape(2, T, 1, 4).clone())
            ro.update_goals(goal[0], goal[1], torch.as_tensor(np.arange(B, dtype=np.int32) % 2, device=dev))
            c, g = ro.cost_and_gradient(knots)
            torch.cuda.synchronize()
            out.append((c.clone(), g.clone(), ro))
        (cf, gf, rf), (cs, gs, rs) = out
        dc = float((cf - cs).abs().max() / cs.abs().max())
        dg = float((gf - gs).abs().max() / gs.abs().max())
        print(f"{robot} sweep={sweep}: fused kernel ran: {rf.fused_available()}  rel cost diff {dc:.2e}  rel grad diff {dg:.2e}  "
              f"cost range [{float(cs.min()):.3g}, {float(cs.max()):.3g}]  self-collision active rows {int((rs.self_dist.view(B, -1).sum(-1) > 0).sum())}/{B}", flush=True)
    del goal
