"""pose-to-pose planning on the reference's other two benchmark robots (every tool frame has a goal): does it run, how often does it succeed"""
import json
import sys
import time

import numpy as np
import torch

from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg
from curobo_amd.types import JointState

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
delta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0  # > 0: goals = start + U(-delta, delta) per joint instead of independent samples
for robot in sys.argv[3:] or ("dual_ur10e", "unitree_g1"):
    try:
        planner = MotionPlanner(MotionPlannerCfg.create(robot=f"{robot}.yml", scene_model="collision_table.yml"))
        t0 = time.perf_counter()
        planner.warmup()
        warm = time.perf_counter() - t0
        torch.manual_seed(3)
        q = planner.sample_configs(2 * n + 20, rejection_ratio=50)
        assert q.shape[0] >= 2 * n, q.shape
        if delta > 0:
            lo, hi = planner.kinematics.kinematics_config.joint_limits_position
            g = torch.minimum(torch.maximum(q[:n] + delta * (2 * torch.rand(n, q.shape[1], device=q.device) - 1), lo + 0.01), hi - 0.01)
            q = torch.cat([q[:n], g], 0)
        ok, ms, why = 0, [], {}
        for i in range(n):
            cur = JointState.from_position(q[i:i + 1].clone(), planner.joint_names)
            goal = planner.compute_kinematics(JointState.from_position(q[n + i:n + i + 1].clone(), planner.joint_names)).tool_poses.as_goal()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = planner.plan_pose(goal, cur, max_attempts=2)
            torch.cuda.synchronize()
            if r is not None and bool(r.success.any()):
                ok += 1
                ms.append(1e3 * (time.perf_counter() - t0))
            else:
                k = "IK found nothing" if r is None else "trajectory optimisation failed"
                why[k] = why.get(k, 0) + 1
        print(json.dumps({"robot": robot, "dof": planner.action_dim, "tool_frames": len(planner.tool_frames), "problems": n, "goal_delta": delta, "success_percent": 100.0 * ok / n,
                          "plan_ms_median": float(np.median(ms)) if ms else None, "plan_ms_max": float(np.max(ms)) if ms else None, "failures": why,
                          "warmup_s": round(warm, 2)}), flush=True)
    except Exception as e:  # noqa: BLE001
        import traceback
        print(json.dumps({"robot": robot, "error": f"{type(e).__name__}: {str(e)[:500]}", "where": traceback.format_exc()[-600:]}), flush=True)
