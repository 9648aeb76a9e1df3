"""which check fails when planning for a multi-frame robot"""
import sys
import numpy as np
import torch
from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg
from curobo_amd.solver import trajopt as T
from curobo_amd.types import JointState

robot = sys.argv[1] if len(sys.argv) > 1 else "dual_ur10e"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
planner = MotionPlanner(MotionPlannerCfg.create(robot=f"{robot}.yml", scene_model="collision_table.yml"))
planner.warmup()
last = {}
orig = T.TrajOptSolver._rank


def spy(self, best, seed_goal, k, passes):
    last["best"] = {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in best.items()}
    last["passes"] = passes
    return orig(self, best, seed_goal, k, passes)


T.TrajOptSolver._rank = spy
torch.manual_seed(3)
q = planner.sample_configs(2 * n + 20, rejection_ratio=50)
delta = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
if delta > 0:  # goals near the starts instead of independent samples
    lo, hi = planner.kinematics.kinematics_config.joint_limits_position
    g = torch.minimum(torch.maximum(q[:n] + delta * (2 * torch.rand(n, q.shape[1], device=q.device) - 1), lo + 0.01), hi - 0.01)
    q = torch.cat([q[:n], g, q[2 * n:]], 0)
for i in range(n):
    cur = JointState.from_position(q[i:i + 1].clone(), planner.joint_names)
    goal = planner.compute_kinematics(JointState.from_position(q[n + i:n + i + 1].clone(), planner.joint_names)).tool_poses.as_goal()
    ok, seeds = planner._ik_seed_configs(goal, 1)
    r = planner.plan_pose(goal, cur, max_attempts=1)
    b = last.get("best")
    f = lambda k: b[k].view(-1).int().tolist()  # noqa: E731
    print(i, "ik ok", ok.view(-1).int().tolist(), "success", None if r is None else r.success.view(-1).int().tolist(), "converged", f("converged"), "limits", f("in_limits"),
          "self", f("no_self_collision"), "scene", f("no_scene_collision"), "interp", f("feasible_interpolated"),
          "pos_err", b["pos_err"].view(-1).cpu().numpy().round(4), "rot_err", b["rot_err"].view(-1).cpu().numpy().round(3), "dt", b["dt"].view(-1).cpu().numpy().round(4),
          "passes", last.get("passes"), flush=True)
    mm = planner.trajopt_solver._solver.metrics_rollout
    Hh = mm.cfg.padded_horizon
    sd = mm.self_dist.view(-1, Hh)
    print("   self_dist (last pass): per-seed max", sd.max(-1).values.cpu().numpy().round(5).tolist(), "points in collision", (sd > 0).sum(-1).tolist(),
          "argmax point", sd.argmax(-1).tolist(), "travel", (mm.position[:, -1] - mm.position[:, 0]).abs().max(-1).values.cpu().numpy().round(2).tolist())
    m = planner.trajopt_solver._solver.metrics_rollout if hasattr(planner.trajopt_solver, "_solver") else None
    if m is not None and i == 0:
        H, Tn = m.pose_pos_dist.shape[-2] if m.pose_pos_dist.ndim > 2 else None, planner.kinematics.kinematics_config.num_pose_links
        print("   per-frame last-point errors:", m.pose_pos_dist.view(-1, m.cfg.padded_horizon, Tn)[:, -1].cpu().numpy().round(4).tolist(),
              m.pose_rot_dist.view(-1, m.cfg.padded_horizon, Tn)[:, -1].cpu().numpy().round(3).tolist())
