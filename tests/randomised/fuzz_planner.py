"""End to end on random worlds: TrajOptSolver.solve_pose (IK -> trajectory optimisation -> finetune -> retime) for feasible random
goals; every reported SUCCESS is verified with the oracle (starts at the start, reaches the pose, inside the joint limits, free
of self and scene collision over the horizon), no result may be non-finite, and the success rate is reported.
    python tests/randomised/fuzz_planner.py [worlds] [seed] [--mesh]

--mesh (end of round 6): every obstacle but the table is given as a triangle mesh (a box of 12 x 4^k triangles, a ball, a torus) in
a store with the sign-consistent gradient -- what ``scene_from_config`` builds -- and the successes are verified against the
oracle's brute force over every triangle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_model  # noqa: E402

from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.solver import TrajOptSolver, TrajOptSolverCfg  # noqa: E402
from curobo_amd.workloads import feasible_goals, start_configuration  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

dev = torch.device("cuda:0")
oracle = Oracle()
MESH = "--mesh" in sys.argv
sys.argv = [a for a in sys.argv if a != "--mesh"]
n_worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
model = load_model("franka")
md = model.as_dict()
kin = KinematicsParams.from_model(model, dev)
start = start_configuration(model)
P = 8
bad = 0
rates = []
for wi in range(n_worlds):
    world = [{"dims": [2.2, 2.2, 0.2], "pose": [0.0, 0.0, -0.12, 1, 0, 0, 0]}]  # a table under the robot
    for _ in range(int(rng.integers(1, 5))):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        r, a = rng.uniform(0.35, 0.75), rng.uniform(0, 2 * np.pi)  # around the robot, not on its base
        world.append({"dims": [float(v) for v in rng.uniform(0.05, 0.3, size=3)],
                      "pose": [float(r * np.cos(a)), float(r * np.sin(a)), float(rng.uniform(0.1, 0.8))] + [float(v) for v in q]})
    arrays = cuboid_scene_arrays([world])
    scene = SceneData.from_arrays(arrays, dev)
    if MESH:
        from test_oracle_mesh import box_shape, sphere_shape, torus_shape

        from curobo_amd.scene import MeshStore
        from oracle.oracle import mesh_scene_arrays

        meshes = []
        for i, o in enumerate(world[1:]):
            kind = int(rng.integers(0, 3))
            if kind == 0:
                v, f = box_shape(o["dims"], int(rng.integers(0, 3)))
            elif kind == 1:
                v, f = sphere_shape(0.5 * float(max(o["dims"])), 12, 24)
            else:
                v, f = torus_shape(0.5 * float(max(o["dims"])) + 0.05, 0.03, 24, 12)
            meshes.append({"name": f"m{i}", "vertices": v, "faces": f, "pose": o["pose"]})
        arrays = {**cuboid_scene_arrays([world[:1]]), **mesh_scene_arrays([meshes])}
        scene = SceneData.from_arrays(cuboid_scene_arrays([world[:1]]), dev, meshes=MeshStore([meshes], dev, gradient_mode=MeshStore.CONSISTENT_GRADIENT))
    try:
        gp, gq = feasible_goals(kin, scene, P)
        solver = TrajOptSolver(kin, scene, P, TrajOptSolverCfg(num_seeds=4))
        res = solver.solve_pose(torch.as_tensor(start), gp, gq)
        torch.cuda.synchronize()
        succ = res.success.cpu().numpy()
        rates.append(float(succ.mean()))
        for name in ("position", "position_error", "rotation_error", "traj_dt"):
            assert torch.isfinite(getattr(res, name)).all(), f"{name} is not finite"
        traj = res.position.cpu().numpy()[succ]
        n = traj.shape[0]
        if n:
            H, D = traj.shape[1:]
            np.testing.assert_allclose(traj[:, 0], np.broadcast_to(start, (n, D)), atol=1e-4)
            chk = oracle.kinematics_forward(traj.reshape(n * H, D), md, horizon=H)
            np.testing.assert_allclose(chk["link_pos"].reshape(n, H, 3)[:, -1], gp.cpu().numpy()[succ], atol=5e-3)
            lo, hi = model.joint_limits_position
            assert (traj >= lo - 1e-3).all() and (traj <= hi + 1e-3).all(), "joint limits"
            s2 = chk["robot_spheres"].reshape(n, H, -1, 4)
            assert (oracle.self_collision(s2, model.sphere_padding, model.collision_pairs, 1.0)["distance"] == 0).all(), "self collision on a success"
            assert (oracle.scene_collision(s2, arrays, 1.0, 0.0)["distance"].sum((1, 2)) == 0).all(), "scene collision on a success"
        s0 = oracle.kinematics_forward(np.asarray(start, np.float32)[None], md)["robot_spheres"].reshape(1, 1, -1, 4)
        start_pen = float(oracle.scene_collision(s0, arrays, 1.0, 0.0)["distance"].sum())
        print(f"world {wi}: {len(world)} {'obstacles (meshes but the table)' if MESH else 'cuboids'}{', START IN COLLISION' if start_pen > 0 else ''}, IK success {float(res.ik_success.float().mean()):.2f}, trajopt success {succ.mean():.2f}, passes {res.finetune_passes}", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"FAILED world {wi}: {type(e).__name__}: {str(e)[:400]}".replace("\n", " | "))
print(f"{n_worlds} worlds, {bad} failed; mean success rate {np.mean(rates) if rates else 0:.2f}")
