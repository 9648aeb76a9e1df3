"""Randomised parity sweep of the fused rollout launch against the kernel sequence (the harness of tests/test_gpu_fused.py on
random worlds and shapes): robots, batch sizes, spline degrees / knots / interpolation steps (horizons 9 .. 65), 0 .. 12
cuboids with random poses (disabled slots among them), an ESDF grid, sweep / speed metric / self / scene on and off.
    python tests/randomised/fuzz_fused.py [cases] [seed]"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fused as T  # noqa: E402

dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def random_world(n):
    out = []
    for i in range(n):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        if rng.random() < 0.3:
            q = np.array([1.0, 0, 0, 0])
        dims = rng.uniform(0.05, 0.6, size=3)
        if rng.random() < 0.2:
            dims[rng.integers(3)] = rng.uniform(0.8, 2.2)  # a slab / a pillar
        pos = rng.uniform([-0.7, -0.7, -0.2], [0.7, 0.7, 1.0])
        o = {"dims": [float(v) for v in dims], "pose": [float(v) for v in pos] + [float(v) for v in q]}
        if rng.random() < 0.15:
            o["enable"] = False
        out.append(o)
    return [out]


bad = 0
ran = 0
n_amb_off = n_amb = n_traj = 0
for case in range(n_cases):
    robot = "franka" if rng.random() < 0.7 else "ur10e"
    degree = int(rng.choice([3, 4, 5]))
    n_knots = int(rng.choice([4, 6, 8, 12, 16]))
    interp = int(rng.choice([1, 2, 3, 4]))
    seeds = int(rng.integers(1, 70))
    n_obs = int(rng.integers(0, 13))
    kw = dict(use_sweep=bool(rng.random() < 0.7), use_self_collision=bool(rng.random() < 0.85), n_knots=n_knots,
              interpolation_steps=interp, bspline_degree=degree)
    kw["use_speed_metric"] = kw["use_sweep"] and bool(rng.random() < 0.7)
    kw["use_scene_collision"] = n_obs > 0 and (not kw["use_self_collision"] or rng.random() < 0.9)
    voxel = bool(rng.random() < 0.2) and kw["use_scene_collision"]
    desc = f"case {case}: {robot} (multi-env worlds in a quarter of the cases) seeds {seeds} degree {degree} knots {n_knots} x {interp} obstacles {n_obs} voxel {voxel} {kw}"
    if not (kw["use_self_collision"] or kw["use_scene_collision"]):
        continue
    try:
        n_env = int(rng.choice([1, 1, 2, 3]))
        world = [random_world(max(n_obs, 1))[0] for _ in range(n_env)]  # several environments: different obstacle sets (and counts)
        if n_env > 1:
            world = [w[: max(1, int(rng.integers(1, len(w) + 1)))] for w in world]
        _, _, knots, _, ro_ref, ro_fused = T._pair(dev, robot=robot, seeds=seeds, world=world, voxel=voxel and n_env == 1, **kw)
        if n_env > 1:
            env = torch.as_tensor(rng.integers(0, n_env, size=seeds).astype(np.int32), device=dev)
            ro_ref.update_env_query_idx(env)
            ro_fused.update_env_query_idx(env)
        if not ro_fused.fused_available():
            continue
        knots = knots * float(rng.uniform(0.3, 1.2))
        try:
            c0, g0, c1, g1 = T._compare(ro_ref, ro_fused, knots, dev)
        except AssertionError as e:
            if "c0.max" in str(e) or str(e) == "":
                continue  # (no cost anywhere: nothing to compare)
            raise
        ran += 1
        # trajectories with a sphere that is stationary up to rounding AND in scene collision: the reference's sweep adds a
        # duplicate of the centre sample iff half_dist > 0 (the known discontinuity; see test_fused_swept_matches_oracle_at_c2_size)
        p = ro_fused.robot_spheres.cpu().numpy()[..., :3]
        stepn = np.linalg.norm(np.diff(p, axis=1), axis=-1)
        still = np.zeros(p.shape[:3], bool)
        still[:, 1:] |= stepn < 1e-5
        still[:, :-1] |= stepn < 1e-5
        sd = ro_ref.scene_dist.cpu().numpy().reshape(p.shape[:3]) if kw["use_scene_collision"] else np.zeros(p.shape[:3], np.float32)
        amb = (still & (sd > 0)).any(axis=(1, 2)) if kw["use_sweep"] else np.zeros(p.shape[0], bool)
        off = np.abs(c1 - c0) > 2e-5 * np.abs(c0) + 1e-3
        goff = (np.abs(g1 - g0) > 1e-3 * np.abs(g0) + 2e-5 * np.abs(g0).max()).any(-1)
        n_amb_off += int((off & amb).sum())
        n_amb += int(amb.sum())
        n_traj += int(amb.size)
        if (off & ~amb).any() or (goff & ~amb).any():
            raise AssertionError(f"{int((off & ~amb).sum())} cost / {int((goff & ~amb).sum())} gradient mismatches on trajectories WITHOUT a resting colliding sphere "
                                 f"(max rel {float((np.abs(c1 - c0) / np.maximum(np.abs(c0), 1.0))[off & ~amb].max()) if (off & ~amb).any() else 0:.2e}); "
                                 f"{int((off & amb).sum())} of {int(amb.sum())} ambiguous trajectories differ")
        w = ro_fused.cfg.scene_collision_weight
        band = (c1 <= 3.001 * c0 + 1e-3 * w) & (c0 <= 3.001 * c1 + 1e-3 * w)
        assert band[amb].all(), "an ambiguous trajectory outside the 3x band"
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAILED", desc)
        print("   ", type(e).__name__, str(e)[:600].replace("\n", " | "))
        if not isinstance(e, AssertionError):
            traceback.print_exc(limit=4)
print(f"{ran} cases compared, {bad} failed; {n_traj} trajectories, {n_amb} with a resting colliding sphere, of which {n_amb_off} differ between the two paths")
