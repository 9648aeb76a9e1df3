"""which check fails on the planning problems that fail (C2 world, the 100 problems of planner_benchmark.py)"""
import sys
import numpy as np
import torch
from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg
from curobo_amd.scene.types import Cuboid, SceneCfg
from curobo_amd.solver import trajopt as T
from curobo_amd.types import JointState
from curobo_amd.workloads import c2_world

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scene = SceneCfg(cuboid=[Cuboid(f"c{i}", list(o["pose"]), dims=list(o["dims"])) for i, o in enumerate(c2_world()[0])])
planner = MotionPlanner(MotionPlannerCfg.create(robot="franka.yml", scene_model=scene, num_trajopt_seeds=seeds))
planner.warmup()
last = {}
orig = T.TrajOptSolver._rank


def spy(self, best, seed_goal, k, passes):
    last["best"] = {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in best.items()}
    return orig(self, best, seed_goal, k, passes)


T.TrajOptSolver._rank = spy
torch.manual_seed(7)
q = planner.sample_configs(2 * n + 50, rejection_ratio=20)
starts, goals = q[:n], q[n:2 * n]
fails = 0
for i in range(n):
    cur = JointState.from_position(starts[i:i + 1].clone(), planner.joint_names)
    goal = planner.compute_kinematics(JointState.from_position(goals[i:i + 1].clone(), planner.joint_names)).tool_poses.as_goal()
    r = planner.plan_pose(goal, cur, max_attempts=1)
    if r is None or not bool(r.success.any()):
        fails += 1
        if r is None:
            print(i, "IK found nothing")
            continue
        b = last["best"]
        f = lambda k: b[k].view(-1).int().tolist()  # noqa: E731
        print(i, "converged", f("converged"), "limits", f("in_limits"), "self", f("no_self_collision"), "scene", f("no_scene_collision"),
              "interp", f("feasible_interpolated"), "pos_err", b["pos_err"].view(-1).cpu().numpy().round(4), "rot_err", b["rot_err"].view(-1).cpu().numpy().round(3),
              "dt", b["dt"].view(-1).cpu().numpy().round(4), flush=True)
print("failures with one attempt:", fails, "of", n, "seeds", seeds)
