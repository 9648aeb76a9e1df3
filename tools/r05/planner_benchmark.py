"""Pose-to-pose planning over random problems, in the shape of the reference's motion-generation benchmark (its problem set,
motionbenchmaker / mpinets, is not in this image: docs/reference/benchmarks.rst:7-17 reports 99.73 % success, plan time mean 38 ms,
median 35 ms, 98 % 81 ms on an RTX 6000 Ada).  Problems: collision-free start configuration -> the tool pose of another
collision-free configuration, Franka, in (a) collision_table.yml and (b) the C2 world (table, pillar, two blocks).
    python tools/r05/planner_benchmark.py [n_problems] [out.json]"""
import json
import sys
import time

import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg
from curobo_amd.scene.types import Cuboid, SceneCfg
from curobo_amd.types import JointState
from curobo_amd.workloads import c2_world

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def c2_scene():
    return SceneCfg(cuboid=[Cuboid(f"c{i}", list(o["pose"]), dims=list(o["dims"])) for i, o in enumerate(c2_world()[0])])


def c2_mesh_scene():
    """the C2 world with the table as a cuboid and the pillar and the blocks as triangle meshes (12 triangles each)"""
    from curobo_amd.scene import box_mesh

    obs = c2_world()[0]
    meshes = {}
    for i, o in enumerate(obs[1:]):
        v, f = box_mesh(o["dims"])
        meshes[f"m{i}"] = {"vertices": v.astype(np.float32), "faces": f.astype(np.int32), "pose": list(o["pose"])}
    return {"cuboid": {"table": {"dims": list(obs[0]["dims"]), "pose": list(obs[0]["pose"])}}, "mesh": meshes}


out = []
cases = [("collision_table.yml", "collision_table.yml"), ("C2 world (table, pillar, two blocks)", c2_scene())]
if os.environ.get("PLANNER_BENCH_MESH", "0") == "1":  # (round 6: the same world with meshes, same problems)
    cases = [cases[1], ("C2 world, pillar and blocks as triangle meshes", c2_mesh_scene())]
for label, scene in cases:
    planner = MotionPlanner(MotionPlannerCfg.create(robot="franka.yml", scene_model=scene))
    planner.warmup()
    torch.manual_seed(7)
    q = planner.sample_configs(2 * n + 50, rejection_ratio=20)
    assert q.shape[0] >= 2 * n, q.shape
    starts, goals = q[:n], q[n:2 * n]
    rec = {"world": label, "problems": n, "total_ms": [], "solve_ms": [], "motion_s": [], "fail": {}}
    ok = 0
    for i in range(n):
        cur = JointState.from_position(starts[i:i + 1].clone(), planner.joint_names)
        goal = planner.compute_kinematics(JointState.from_position(goals[i:i + 1].clone(), planner.joint_names)).tool_poses.as_goal()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = planner.plan_pose(goal, cur)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if r is not None and bool(r.success.any()):
            ok += 1
            rec["total_ms"].append(1e3 * dt)
            rec["solve_ms"].append(1e3 * float(r.solve_time))
            mt = getattr(r, "motion_time", None)
            if mt is not None:
                rec["motion_s"].append(float(torch.as_tensor(mt).reshape(-1)[0]))
        else:
            why = "IK found nothing" if r is None else str(getattr(r, "status", "trajectory optimisation failed"))
            rec["fail"][why] = rec["fail"].get(why, 0) + 1
            rec.setdefault("fail_ms", []).append(1e3 * dt)
    t = np.asarray(rec["total_ms"])
    summary = {"world": label, "problems": n, "success_percent": 100.0 * ok / n,
               "plan_ms": {"mean": float(t.mean()), "std": float(t.std()), "median": float(np.median(t)), "p75": float(np.percentile(t, 75)),
                           "p98": float(np.percentile(t, 98)), "max": float(t.max())},
               "solve_ms_mean": float(np.mean(rec["solve_ms"])), "motion_s_mean": float(np.mean(rec["motion_s"])) if rec["motion_s"] else None,
               "failures": rec["fail"], "failed_attempt_ms_mean": float(np.mean(rec["fail_ms"])) if rec.get("fail_ms") else None,
               "published_rtx6000ada": {"success_percent": 99.73, "plan_ms_mean": 38.0, "plan_ms_median": 35.0, "plan_ms_p98": 81.0,
                                        "note": "other problem set (motionbenchmaker + mpinets, 2600 problems)"}}
    out.append(summary)
    print(json.dumps(summary), flush=True)
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        json.dump(out, fh, indent=1)
