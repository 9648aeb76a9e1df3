"""BatchMotionPlanner over random problems: 16 problems per pass in one world (C2 shapes), 10 passes; success per problem with one
and with three attempts, wall time per pass.    python tools/r05/batch_planner_benchmark.py [out.json]"""
import json
import sys
import time

import numpy as np
import torch

from curobo_amd.motion_planner import BatchMotionPlanner, MotionPlannerCfg
from curobo_amd.scene.types import Cuboid, SceneCfg
from curobo_amd.types import JointState
from curobo_amd.workloads import c2_world

B, passes = 16, 10
scene = SceneCfg(cuboid=[Cuboid(f"c{i}", list(o["pose"]), dims=list(o["dims"])) for i, o in enumerate(c2_world()[0])])
out = []
for attempts in (1, 3):
    planner = BatchMotionPlanner(MotionPlannerCfg.create(robot="franka.yml", scene_model=scene, max_batch_size=B))
    planner.warmup()
    torch.manual_seed(11)
    q = planner.sample_configs(2 * B * passes + 100, rejection_ratio=20)
    assert q.shape[0] >= 2 * B * passes
    ok, ms = [], []
    for p in range(passes):
        s = q[2 * B * p: 2 * B * p + B].clone()
        g = q[2 * B * p + B: 2 * B * (p + 1)].clone()
        goal = planner.compute_kinematics(JointState.from_position(g, planner.joint_names)).tool_poses.as_goal()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = planner.plan_pose(goal, JointState.from_position(s, planner.joint_names), max_attempts=attempts)
        torch.cuda.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
        ok.append(float(r.success.any(dim=-1).float().mean()) if r is not None else 0.0)
    rec = {"batch": B, "passes": passes, "max_attempts": attempts, "success_percent": 100 * float(np.mean(ok)), "per_pass_success": ok,
           "ms_per_pass": {"mean": float(np.mean(ms)), "median": float(np.median(ms)), "max": float(np.max(ms))},
           "ms_per_problem": float(np.mean(ms)) / B}
    out.append(rec)
    print(json.dumps(rec), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
