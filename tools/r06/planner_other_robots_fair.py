"""pose-to-pose planning on the reference's other two benchmark robots with problems that HAVE a solution: start and goal are both
collision-free configurations (the goal either an independent sample or a collision-free configuration within +-delta of the start
per joint), the planner's default number of attempts.  tools/r05/planner_other_robots.py drew `start + U(-delta, delta)` without
checking it and gave up after two attempts: part of its 60 % / 75 % was goals inside the robot or the table.
    python tools/r06/planner_other_robots_fair.py [problems] [delta, 0 = independent samples] [robot ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg  # noqa: E402
from curobo_amd.types import JointState  # noqa: E402

from curobo_amd.solver import trajopt as T  # noqa: E402

last = {}
_orig_rank = T.TrajOptSolver._rank


def _spy(self, best, seed_goal, k, passes):  # (DIAG=1: which check failed on every seed of a failed problem)
    last["best"] = {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in best.items()}
    return _orig_rank(self, best, seed_goal, k, passes)


if os.environ.get("DIAG") == "1":
    T.TrajOptSolver._rank = _spy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
delta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
for robot in sys.argv[3:] or ("dual_ur10e", "unitree_g1"):
    planner = MotionPlanner(MotionPlannerCfg.create(robot=f"{robot}.yml", scene_model="collision_table.yml"))
    planner.warmup()
    torch.manual_seed(3)
    q = planner.sample_configs(2 * n + 20, rejection_ratio=50)
    assert q.shape[0] >= 2 * n, q.shape
    starts, goals = q[:n], q[n:2 * n].clone()
    unchecked_in_collision = None
    if delta > 0:
        chk = planner.trajopt_solver._sample_checker[1]
        lo, hi = planner.kinematics.kinematics_config.joint_limits_position
        first = torch.minimum(torch.maximum(starts + delta * (2 * torch.rand(n, q.shape[1], device=q.device) - 1), lo + 0.01), hi - 0.01)
        unchecked_in_collision = int((~chk.validate(first.unsqueeze(1)).view(-1)).sum())
        for i in range(n):
            for _ in range(200):
                g = torch.minimum(torch.maximum(starts[i:i + 1] + delta * (2 * torch.rand(64, q.shape[1], device=q.device) - 1), lo + 0.01), hi - 0.01)
                ok = chk.validate(g.unsqueeze(1)).view(-1)
                if bool(ok.any()):
                    goals[i] = g[ok][0]
                    break
            else:
                raise RuntimeError(f"no collision-free goal near start {i}")
    ok_n, ms, why, attempts = 0, [], {}, []
    for i in range(n):
        cur = JointState.from_position(starts[i:i + 1].clone(), planner.joint_names)
        goal = planner.compute_kinematics(JointState.from_position(goals[i:i + 1].clone(), planner.joint_names)).tool_poses.as_goal()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = planner.plan_pose(goal, cur)
        torch.cuda.synchronize()
        if r is not None and bool(r.success.any()):
            ok_n += 1
            ms.append(1e3 * (time.perf_counter() - t0))
        else:
            k = "IK found nothing" if r is None else "trajectory optimisation failed"
            why[k] = why.get(k, 0) + 1
            if r is not None and "best" in last:
                b = last["best"]
                f = lambda kk: b[kk].view(-1).int().tolist()  # noqa: E731
                print("   problem", i, "last attempt: converged", f("converged"), "limits", f("in_limits"), "self", f("no_self_collision"), "scene", f("no_scene_collision"),
                      "interp", f("feasible_interpolated"), "pos_err", b["pos_err"].view(-1).cpu().numpy().round(4).tolist(),
                      "rot_err", b["rot_err"].view(-1).cpu().numpy().round(3).tolist(), "dt", b["dt"].view(-1).cpu().numpy().round(3).tolist(), flush=True)
                sol = planner.trajopt_solver.solver
                m = sol.metrics_rollout  # (its buffers hold the last pass)
                sd = m.self_dist.view(b["pos_err"].numel(), -1)
                lo_, hi_ = sol.kin.joint_limits_position[0], sol.kin.joint_limits_position[1]
                qq = b["position"]
                over = ((qq < lo_ - 1e-4) | (qq > hi_ + 1e-4)).any(-1)
                vb, ab, jb = m._v_b, m._a_b, m._j_b
                print("      points in self collision per seed", (sd > 0).sum(-1).tolist(), "first / last point", (sd[:, 0] > 0).int().tolist(), (sd[:, -1] > 0).int().tolist(),
                      "| points outside the position limits", over.sum(-1).tolist(),
                      "| max |v| / limit", float((b["velocity"].abs() / vb[1]).max()), "|a|", float((b["acceleration"].abs() / ab[1]).max()),
                      "|j|", float((b["jerk"].abs() / jb[1]).max()), flush=True)
    print(json.dumps({"robot": robot, "dof": planner.action_dim, "tool_frames": len(planner.tool_frames), "problems": n, "goal_delta": delta,
                      "goals_of_the_unchecked_generator_in_collision": unchecked_in_collision, "success_percent": 100.0 * ok_n / n,
                      "plan_ms_median": float(np.median(ms)) if ms else None, "plan_ms_max": float(np.max(ms)) if ms else None, "failures": why}), flush=True)
