"""Success rate / solve time of TrajOptSolver when every seed ends in the single best IK solution
(num_ik_goals=1) vs when seed s ends in the s-th best IK solution (the reference's seeding,
solver_trajopt.py:390-420).  Run on the GPU box: python tools/trajopt_goal_diversity.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from curobo_amd.robot import RobotModel
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.scene import SceneData, cuboid_scene_arrays
    from curobo_amd.solver import TrajOptSolver, TrajOptSolverCfg
    from curobo_amd.workloads import c1_world, c2_world, feasible_goals, start_configuration

    dev = torch.device("cuda:0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = RobotModel.load_npz(os.path.join(root, "curobo_amd", "content", "robot", "franka.npz"))
    kin = KinematicsParams.from_model(model, dev)
    out = {}
    P = int(os.environ.get("P", 32))
    for wname, world in (("c1", c1_world()), ("c2", c2_world())):
        scene = SceneData.from_arrays(cuboid_scene_arrays(world), dev)
        gp, gq = feasible_goals(kin, scene, P)  # FK of collision-free configurations (reference benchmark protocol)
        start = torch.as_tensor(start_configuration(model))
        for S, K in ((4, 1), (4, 4), (8, 1), (8, 8)):
            slv = TrajOptSolver(kin, scene, P, TrajOptSolverCfg(num_seeds=S, num_ik_goals=K))
            res = slv.solve_pose(start, gp, gq)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = slv.solve_pose(start, gp, gq)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[f"{wname}_S{S}_K{K}"] = {"success": round(float(res.success.float().mean()), 3),
                                        "ik_success": round(float(res.ik_success.float().mean()), 3),
                                        "ms": round(dt * 1e3, 1)}
            print(wname, S, K, out[f"{wname}_S{S}_K{K}"], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/trajopt_goal_diversity.json", "w"), indent=1)


if __name__ == "__main__":
    main()
