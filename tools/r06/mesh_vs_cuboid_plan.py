"""one planning problem around a pillar: the pillar (and a ball) as triangle meshes vs as the analytic primitives of the cuboid
store; per seed: success, and what the oracle says about the winner (scene / self collision, goal error)"""
import os
import sys

import numpy as np
import torch

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(root, "tests"))
sys.path.insert(0, root)
from test_oracle_mesh import box_shape, sphere_shape  # noqa: E402

from curobo_amd.motion_planner import MotionPlanner, MotionPlannerCfg  # noqa: E402
from curobo_amd.scene import cuboid_scene_arrays  # noqa: E402
from curobo_amd.types import JointState  # noqa: E402
from oracle.oracle import Oracle, mesh_scene_arrays  # noqa: E402

orc = Oracle()
vb, fb = box_shape([0.16, 0.16, 0.7], 2)
vs, fs = sphere_shape(0.12)
table = {"dims": [2.0, 2.0, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]}
pillar_pose, ball_pose = [0.5, 0.0, 0.35, 1, 0, 0, 0], [0.0, 0.55, 0.9, 0.9238795, 0, 0.3826834, 0]
meshes = {"pillar": {"vertices": vb, "faces": fb, "pose": pillar_pose}, "ball": {"vertices": vs, "faces": fs, "pose": ball_pose}}
worlds = {
    "mesh": {"cuboid": {"table": table}, "mesh": meshes},
    "cuboid": {"cuboid": {"table": table, "pillar": {"dims": [0.16, 0.16, 0.7], "pose": pillar_pose}},
               "sphere": {"ball": {"radius": 0.12, "pose": ball_pose}}},
}
seeds = int(os.environ.get("SEEDS", "4"))
for name, world in worlds.items():
    config = MotionPlannerCfg.create(robot="franka.yml", scene_model=world, num_ik_seeds=32, num_trajopt_seeds=seeds)
    planner = MotionPlanner(config)
    model = config.trajopt_solver_config.kinematics.model
    q0 = torch.tensor([[-0.9, 0.3, 0.0, -1.9, 0.0, 2.2, 0.8]], device="cuda")
    cur = JointState.from_position(q0, planner.joint_names)
    goal_js = cur.clone()
    goal_js.position[0, 0] = 0.9
    goal = planner.compute_kinematics(goal_js).tool_poses.as_goal()
    for attempt in range(3):
        ok, seed_config = planner._ik_seed_configs(goal, 1, cur)
        if int(ok.sum()) < ok.shape[1]:
            seed_config = torch.where(ok.unsqueeze(-1), seed_config, seed_config[ok][0:1].view(1, 1, -1))
        r = planner.trajopt_solver.solver.solve_pose if False else None
        res = planner.trajopt_solver.solve_pose(goal, cur, seed_config=seed_config, use_implicit_goal=True, finetune_attempts=1, finetune_dt_scale=0.55)
        sol = planner.trajopt_solver.solver
        mr = sol.metrics_rollout
        print("   per-seed success per pass", [t["success"].view(-1).int().tolist() for t in sol.last_pass_trace[1:]],
              "metrics scene cost per seed", np.round(mr.scene_dist.view(seeds, -1).sum(-1).cpu().numpy(), 3).tolist(),
              "self", np.round(mr.self_dist.view(seeds, -1).sum(-1).cpu().numpy(), 3).tolist(), flush=True)
        ro = sol.rollout
        nls = ro.batch_size // seeds
        kn = sol.optimizer.best_action if hasattr(sol.optimizer, "best_action") else None
        opts = getattr(sol.optimizer, "opts", [sol.optimizer])
        for o in opts:
            x = o.best_action.reshape(-1, o.best_action.shape[-1])
            xx = x.repeat_interleave(ro.batch_size // x.shape[0], 0) if x.shape[0] != ro.batch_size else x
            c, g = ro.cost_and_gradient(xx.contiguous())
            torch.cuda.synchronize()
            print("   optimiser rollout at best_action: cost", np.round(c.view(x.shape[0], -1)[:, 0].cpu().numpy(), 1).tolist(),
                  "scene", np.round(ro.scene_dist.view(ro.batch_size, -1).sum(-1).view(x.shape[0], -1)[:, 0].cpu().numpy(), 2).tolist(),
                  "|grad|", np.round(g.view(x.shape[0], -1, g.shape[-1])[:, 0].norm(dim=-1).cpu().numpy(), 2).tolist(),
                  "|scene_grad|", np.round(ro.scene_grad.view(ro.batch_size, -1).norm(dim=-1).view(x.shape[0], -1)[:, 0].cpu().numpy(), 2).tolist(),
                  "iterations", getattr(o, "iterations_run", None), flush=True)
        traj = res.js_solution.position[0].cpu().numpy()
        H = traj.shape[1]
        sph = orc.kinematics_forward(traj.reshape(-1, 7), model.as_dict(), horizon=H)["robot_spheres"].reshape(1, H, -1, 4)
        arrays = {**cuboid_scene_arrays([[table]]), **mesh_scene_arrays([[dict(m, name=k) for k, m in meshes.items()]])}
        d = orc.scene_collision(sph, arrays, 1.0, 0.0)["distance"]
        sc = orc.self_collision(sph, model.sphere_padding, model.collision_pairs, 1.0)["distance"]
        print(name, "attempt", attempt, "ik ok", int(ok.sum()), "success", bool(res.success[0, 0]), "pos err", float(res.position_error[0, 0]),
              "dt", float(res.js_solution.dt[0, 0]), "oracle scene", float(d.sum()), "steps in collision", int((d.sum(-1) > 0).sum()),
              "self", float(sc.sum()), "seed goal configs", np.round(seed_config[0].cpu().numpy()[:, 0], 2).tolist(), flush=True)
