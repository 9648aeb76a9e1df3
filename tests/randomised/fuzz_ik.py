"""Collision-free IK end to end on random worlds and robots: IKSolver.solve_pose for feasible random goals (exit_early on and
off, 4-64 seeds); every reported SUCCESS is verified with the oracle (pose reached, inside the limits, free of self and
scene collision).   python tests/randomised/fuzz_ik.py [worlds] [seed]"""
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
from curobo_amd.solver import IKSolver, IKSolverCfg  # noqa: E402
from curobo_amd.workloads import feasible_goals  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

dev = torch.device("cuda:0")
oracle = Oracle()
n_worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
rates = []
for wi in range(n_worlds):
    robot = "franka" if rng.random() < 0.7 else "ur10e"
    model = load_model(robot)
    md = model.as_dict()
    kin = KinematicsParams.from_model(model, dev)
    world = [{"dims": [2.2, 2.2, 0.2], "pose": [0.0, 0.0, -0.12, 1, 0, 0, 0]}]
    for _ in range(int(rng.integers(0, 7))):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        r, a = rng.uniform(0.3, 0.8), rng.uniform(0, 2 * np.pi)
        world.append({"dims": [float(v) for v in rng.uniform(0.05, 0.4, size=3)],
                      "pose": [float(r * np.cos(a)), float(r * np.sin(a)), float(rng.uniform(0.05, 0.9))] + [float(v) for v in q]})
    arrays = cuboid_scene_arrays([world])
    scene = SceneData.from_arrays(arrays, dev)
    P, S, early = int(rng.choice([1, 7, 32, 100])), int(rng.choice([4, 16, 64])), bool(rng.random() < 0.5)
    try:
        try:
            gp, gq = feasible_goals(kin, scene, P)
        except RuntimeError as e:  # (rejection sampling at the reference's ratio of 10 can run short for one problem in a crowded world)
            print(f"world {wi}: skipped ({e})")
            continue
        solver = IKSolver(kin, scene, P, IKSolverCfg(num_seeds=S, exit_early=early))
        res = solver.solve_pose(gp, gq)
        torch.cuda.synchronize()
        succ = res.success.cpu().numpy().reshape(P)
        rates.append(float(succ.mean()))
        assert torch.isfinite(res.solution).all() and torch.isfinite(res.position_error).all(), "non-finite result"
        qs = res.solution.cpu().numpy().reshape(P, -1)[succ]
        if len(qs):
            chk = oracle.kinematics_forward(qs, md)
            np.testing.assert_allclose(chk["link_pos"][:, 0], gp.cpu().numpy()[succ], atol=5e-3)
            dotq = np.abs((chk["link_quat"][:, 0] * gq.cpu().numpy()[succ]).sum(-1))
            assert (2 * np.arccos(np.clip(dotq, 0, 1)) < 0.05).all(), "orientation"
            lo, hi = model.joint_limits_position
            assert (qs >= lo - 1e-4).all() and (qs <= hi + 1e-4).all(), "joint limits"
            s2 = chk["robot_spheres"].reshape(len(qs), 1, -1, 4)
            assert (oracle.self_collision(s2, model.sphere_padding, model.collision_pairs, 1.0)["distance"] == 0).all(), "self collision on a success"
            assert (oracle.scene_collision(s2, arrays, 1.0, 0.0)["distance"].sum((1, 2)) == 0).all(), "scene collision on a success"
        print(f"world {wi}: {robot}, {len(world)} cuboids, {P} problems x {S} seeds, exit_early {early}: success {succ.mean():.2f}, L-BFGS ran {solver.optimizer_ran}", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"FAILED world {wi} ({robot}, P {P}, S {S}, exit_early {early}): {type(e).__name__}: {str(e)[:400]}".replace("\n", " | "))
print(f"{n_worlds} worlds, {bad} failed; mean success rate {np.mean(rates) if rates else 0:.2f}")
