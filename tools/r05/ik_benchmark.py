"""The reference's IK benchmark protocol (benchmark/ik_benchmark.py:53-165) over this package's ``InverseKinematics`` front end, on
the three robots it reports (docs/reference/benchmarks.rst:62-72): batch 100, goals = FK of collision-free samples, `IK` = no
collision terms with 2 seeds, `collision-free IK` = self collision + collision_table.yml with 8 (Franka) / 16 seeds, exit_early on,
three warm-up solves then the mean over five goal sets of the wall time of ``solve_pose`` (host clock around a synchronised call).
Differences: the packaged robot models keep their locked joints and collision links in the `IK` case (the reference strips both
there); the seed-solver seed counts follow the reference's (32, 128 for the G1).
    python tools/r05/ik_benchmark.py [out.json]"""
import json
import sys
import time

import numpy as np
import torch

from curobo_amd.solver.inverse_kinematics import InverseKinematics, InverseKinematicsCfg
from curobo_amd.types import JointState

PUBLISHED_MS = {"franka": (2.601, 2.726), "dual_ur10e": (6.058, 15.64), "unitree_g1": (31.39, 526.9)}


def run(robot: str, collision_free: bool, batch: int = 100):
    g1 = robot == "unitree_g1"
    seeds = (16 if robot in ("unitree_g1", "dual_ur10e") else 8) if collision_free else 2
    cfg = InverseKinematicsCfg.create(
        robot=f"{robot}.yml", scene_model="collision_table.yml" if collision_free else None, num_seeds=seeds, position_tolerance=0.005,
        optimizer_collision_activation_distance=0.0025, self_collision_check=collision_free, use_cuda_graph=True,
        seed_solver_num_seeds=128 if g1 else max(32, 2 * seeds), max_batch_size=batch,
        override_iters_for_multi_link_ik=240 if g1 else None)
    ik = InverseKinematics(cfg)
    torch.manual_seed(2)
    sets, ratio = [], 10
    for _ in range(5):
        q = ik.sample_configs(batch, rejection_ratio=ratio)
        while q.shape[0] < batch:
            ratio = int(1.2 * ratio) + 1
            if ratio > 400:
                raise RuntimeError("rejection ratio too high")
            q = ik.sample_configs(batch, rejection_ratio=ratio)
        sets.append(q[:batch].contiguous())
    goal = lambda q: ik.compute_kinematics(JointState.from_position(q)).tool_poses.as_goal()  # noqa: E731
    ik.config.exit_early = False
    for _ in range(3):
        ik.reset_seed()
        ik.solve_pose(goal(sets[0]))
    ik.config.exit_early = True
    times, succ, perr, rerr = [], [], [], []
    for q in sets:
        ik.reset_seed()
        g = goal(q)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = ik.solve_pose(g)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        ok = r.success.view(-1)
        succ.append(100.0 * float(ok.float().mean()))
        if ok.any():
            perr.append(float(np.percentile(r.position_error.view(-1)[ok].cpu().numpy(), 90)))
            rerr.append(float(np.percentile(r.rotation_error.view(-1)[ok].cpu().numpy(), 90)))
    return {"robot": robot, "collision_free": collision_free, "batch": batch, "num_seeds": seeds, "tool_frames": len(ik.tool_frames), "dof": ik.dof,
            "ms": 1e3 * float(np.mean(times)), "ms_each": [round(1e3 * t, 3) for t in times], "success_percent": float(np.mean(succ)),
            "position_error_p90_mm": 1e3 * float(np.mean(perr)) if perr else None, "rotation_error_p90_deg": float(np.degrees(np.mean(rerr))) if rerr else None,
            "published_ms_nvidia": PUBLISHED_MS[robot][1 if collision_free else 0],
            "solves_per_s": batch / float(np.mean(times))}


if __name__ == "__main__":
    out = []
    for robot in ("franka", "dual_ur10e", "unitree_g1"):
        for cfree in (False, True):
            try:
                out.append(run(robot, cfree))
            except Exception as e:  # noqa: BLE001
                out.append({"robot": robot, "collision_free": cfree, "error": f"{type(e).__name__}: {str(e)[:400]}"})
            print(json.dumps(out[-1]), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(out, fh, indent=1)
