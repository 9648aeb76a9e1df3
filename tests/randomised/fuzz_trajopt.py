"""Randomised parity sweep of the fused trajopt launch (pose + c-space STATE [+ torque limits] + self + scene) against the
kernel sequence: random worlds (rotated cuboids), goals, spline shapes, dt, implicit goal state, sweep, torque limits,
non-terminal pose factor.   python tests/randomised/fuzz_trajopt.py [cases] [seed]"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_model, sample_q  # noqa: E402

from curobo_amd.robot.kinematics_params import KinematicsParams  # noqa: E402
from curobo_amd.rollout import TrajOptRollout, TrajOptRolloutCfg  # noqa: E402
from curobo_amd.scene import SceneData, cuboid_scene_arrays  # noqa: E402
from curobo_amd.workloads import seed_knots, start_configuration  # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
model = load_model("franka")
kin = KinematicsParams.from_model(model, dev)
start = torch.as_tensor(start_configuration(model), device=dev)


def random_world(n):
    out = []
    for _ in range(n):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        out.append({"dims": [float(v) for v in rng.uniform(0.05, 0.5, size=3)],
                    "pose": [float(v) for v in rng.uniform([-0.7, -0.7, -0.2], [0.7, 0.7, 1.0])] + [float(v) for v in q]})
    return [out]


bad = ran = skipped = 0
for case in range(n_cases):
    kw = dict(n_knots=int(rng.choice([6, 8, 12, 16])), interpolation_steps=int(rng.choice([1, 2, 3])), traj_dt=float(rng.uniform(0.02, 0.15)),
              use_sweep=bool(rng.random() < 0.6), non_terminal_pose_factor=float(rng.choice([0.0, 0.0, 0.3])))
    kw["use_speed_metric"] = kw["use_sweep"] and bool(rng.random() < 0.7)
    torque = bool(rng.random() < 0.35)
    if torque:
        kw.update(use_torque_limits=True, effort_limit=[float(v) for v in rng.uniform(1.0, 40.0, size=kin.num_dof)], overlap_dynamics=bool(rng.random() < 0.5))
    B = int(rng.integers(1, 40))
    n_goal = int(rng.integers(1, 4))
    implicit = bool(rng.random() < 0.5)
    desc = f"case {case}: B {B} goals {n_goal} implicit {implicit} {kw}"
    try:
        scene = SceneData.from_arrays(cuboid_scene_arrays(random_world(int(rng.integers(1, 9)))), dev)
        knots = torch.as_tensor(seed_knots(model, B, kw["n_knots"], seed=int(rng.integers(1000)), spread=float(rng.uniform(0.2, 0.8))), device=dev)
        gpos = torch.as_tensor(rng.normal(size=(n_goal, 1, 1, 3)).astype(np.float32) * 0.4, device=dev)
        gq = rng.normal(size=(n_goal, 1, 1, 4)).astype(np.float32)
        gq /= np.linalg.norm(gq, axis=-1, keepdims=True)
        idx = torch.as_tensor(rng.integers(0, n_goal, size=B).astype(np.int32), device=dev)
        goal_q = torch.as_tensor(sample_q(model, n_goal, seed=int(rng.integers(1000)), scale=0.5), device=dev)
        ros = []
        for fused in (False, True):
            ro = TrajOptRollout(kin, scene, B, TrajOptRolloutCfg(use_fused=fused, **kw))
            ro.update_start_state(start)
            ro.update_goals(gpos, torch.as_tensor(gq, device=dev), idx)
            if implicit:
                ro.update_goal_state(goal_q, idx)
            ros.append(ro)
        ref, fz = ros
        if not fz.fused_available():
            skipped += 1
            continue
        c1, g1 = [t.clone() for t in fz.cost_and_gradient_fused(knots, with_metrics=True)]
        c0 = ref.evaluate_action(knots, with_gradient=True).clone()
        g0 = ref.grad_knots.view(B, -1).clone()
        torch.cuda.synchronize()
        ran += 1
        torch.testing.assert_close(fz.position, ref.position, rtol=0, atol=2e-6)
        torch.testing.assert_close(fz.robot_spheres, ref.robot_spheres, rtol=0, atol=2e-6)
        torch.testing.assert_close(fz.pose_cost, ref.pose_cost, rtol=2e-4, atol=2e-5 * float(ref.pose_cost.abs().max()) + 1e-12)
        torch.testing.assert_close(fz.cspace_cost, ref.cspace_cost, rtol=2e-4, atol=2e-5 * float(ref.cspace_cost.abs().max()) + 1e-12)
        # trajectories without a sphere that is stationary up to rounding and in scene collision (the sweep's discontinuity)
        p = ref.robot_spheres[..., :3]
        stepn = (p[:, 1:] - p[:, :-1]).norm(dim=-1)
        still = torch.zeros(p.shape[:3], dtype=torch.bool, device=dev)
        still[:, 1:] |= stepn < 1e-5
        still[:, :-1] |= stepn < 1e-5
        amb = (still & (ref.scene_dist.view(p.shape[:3]) > 0)).any(-1).any(-1) if kw["use_sweep"] else torch.zeros(B, dtype=torch.bool, device=dev)
        ok = ~amb
        if ok.any():
            torch.testing.assert_close(c1[ok], c0[ok], rtol=3e-4, atol=1e-1)
            torch.testing.assert_close(g1[ok], g0[ok], rtol=3e-3, atol=1e-4 * float(g0.abs().max()))
        w = 3.001
        band = (c1 <= w * c0 + 1e-3 * fz.cfg.scene_collision_weight) & (c0 <= w * c1 + 1e-3 * fz.cfg.scene_collision_weight)
        assert bool(band.all()), "a trajectory outside the 3x band"
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAILED", desc)
        print("   ", type(e).__name__, str(e)[:700].replace("\n", " | "))
        if not isinstance(e, AssertionError):
            traceback.print_exc(limit=4)
print(f"{ran} cases compared, {skipped} without a fused form, {bad} failed")
