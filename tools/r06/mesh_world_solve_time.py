"""planner solve time around a pillar: the pillar as a cuboid (fused rollout launch) against the pillar as a mesh (kernel
sequence + mesh launch), same problem, median of 20 plans after warm-up"""
import os
import sys
import time

import numpy as np
import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(root, "tests"))
sys.path.insert(0, root)
from test_oracle_mesh import box_shape  # noqa: E402

from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg  # noqa: E402
from curobo_amd.types import JointState  # noqa: E402

vb, fb = box_shape([0.16, 0.16, 0.7], int(os.environ.get("SUBDIV", "2")))
table = {"dims": [2.0, 2.0, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]}
pose = [0.5, 0.0, 0.35, 1, 0, 0, 0]
worlds = {"cuboid": {"cuboid": {"table": table, "pillar": {"dims": [0.16, 0.16, 0.7], "pose": pose}}},
          "mesh": {"cuboid": {"table": table}, "mesh": {"pillar": {"vertices": vb, "faces": fb, "pose": pose}}}}
for seeds in (4, 12):
    for name, world in worlds.items():
        if os.environ.get("ONLY_MESH") and (name != "mesh" or seeds != 4):
            continue
        config = MotionPlannerCfg.create(robot="franka.yml", scene_model=world, num_ik_seeds=32, num_trajopt_seeds=seeds)
        planner = MotionPlanner(config)
        q0 = torch.tensor([[-0.9, 0.3, 0.0, -1.9, 0.0, 2.2, 0.8]], device="cuda")
        cur = JointState.from_position(q0, planner.joint_names)
        g = cur.clone()
        g.position[0, 0] = 0.9
        goal = planner.compute_kinematics(g).tool_poses.as_goal()
        t, s, ok = [], [], 0
        for i in range(25):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = planner.plan_pose(goal, cur, max_attempts=1)
            torch.cuda.synchronize()
            if i >= 5:
                t.append(time.perf_counter() - t0)
                ok += int(res is not None and bool(res.success[0, 0]))
                s.append(res.solve_time if res is not None else float("nan"))
        print(f"{name:7s} trajopt seeds {seeds:2d}  triangles {len(fb) if name == 'mesh' else 0:5d}  plan {1e3 * np.median(t):6.2f} ms  trajopt solve {1e3 * np.nanmedian(s):6.2f} ms  success {ok}/20",
              flush=True)
