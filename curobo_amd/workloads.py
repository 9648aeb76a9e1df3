"""Synthetic, deterministic workloads of the BASELINE configs.

C2 "Franka Panda trajopt: 256 seeds x 32-step horizon, sphere+cuboid world": the cuboid world of
the reference's kernel benchmark (``benchmark/cost_gradient_benchmark.py:486-497``: table
[2.2,2.2,0.2]@[0,0,-0.1] and pillar [0.1,0.1,1.5]@[0.45,0,0.3]) plus two more cuboids.  The
reference turns sphere primitives into meshes (``geom/types.py:1104-1124``), so cuboid
equivalents are used for result parity.
"""

from __future__ import annotations

import math
from typing import Dict, List

import numpy as np


def c2_world() -> List[List[Dict]]:
    c, s = math.cos(0.3), math.sin(0.3)
    return [[
        {"dims": [2.2, 2.2, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]},
        {"dims": [0.1, 0.1, 1.5], "pose": [0.45, 0.0, 0.3, 1, 0, 0, 0]},
        {"dims": [0.3, 0.3, 0.3], "pose": [0.2, 0.55, 0.45, c, 0, 0, s]},
        {"dims": [0.25, 0.25, 0.25], "pose": [-0.2, -0.5, 0.7, 1, 0, 0, 0]},
    ]]


def seed_knots(model, num_seeds: int, n_knots: int, seed: int = 2, seed_offset: int = 0,
               spread: float = 0.35) -> np.ndarray:
    """Deterministic per-seed knot sets: a straight line from the start configuration to a
    per-seed goal plus a smooth per-seed perturbation.  Seed ``i`` depends only on its GLOBAL
    index ``seed_offset + i`` so any sharding of the seed axis sees identical seeds."""
    lo, hi = model.joint_limits_position
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    D = model.num_dof
    out = np.zeros((num_seeds, n_knots, D), np.float32)
    t = np.linspace(0.0, 1.0, n_knots)[:, None]
    start = start_configuration(model)
    for i in range(num_seeds):
        rng = np.random.default_rng([seed, seed_offset + i])
        goal = mid + half * 0.8 * rng.uniform(-1, 1, size=D)
        bump = rng.normal(size=(1, D)) * spread * half * np.sin(np.pi * t)
        out[i] = (start[None] * (1 - t) + goal[None] * t + bump).astype(np.float32)
    return np.clip(out, lo + 1e-3, hi - 1e-3).astype(np.float32)


def start_configuration(model) -> np.ndarray:
    q = np.asarray(model.cspace.get("default_joint_position", [0.0] * model.num_dof), dtype=np.float64)
    if q.shape[0] != model.num_dof:
        q = 0.5 * (model.joint_limits_position[0] + model.joint_limits_position[1])
    return q.astype(np.float32)


def c1_world() -> List[List[Dict]]:
    """C1 "4-primitive world" for IK: the reference's table (``collision_table.yml``: dims
    [2.0, 2.0, 0.2] under the base) plus three boxes in the workspace."""
    return [[
        {"dims": [2.0, 2.0, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]},
        {"dims": [0.2, 0.2, 0.6], "pose": [0.55, 0.35, 0.3, 1, 0, 0, 0]},
        {"dims": [0.2, 0.2, 0.6], "pose": [0.55, -0.35, 0.3, 1, 0, 0, 0]},
        {"dims": [0.3, 0.3, 0.05], "pose": [-0.5, 0.0, 0.6, 1, 0, 0, 0]},
    ]]


def reachable_goals(kin, num: int, seed: int = 0, scale: float = 0.8):
    """Goal poses (position [num,3], quaternion wxyz [num,4]) = FK of random joint samples, so that
    every goal is kinematically reachable (the reference's ik_benchmark.py:88-94 protocol)."""
    import torch

    from .kinematics import Kinematics, KinematicsCfg

    g = torch.Generator().manual_seed(seed)
    lo, hi = kin.joint_limits_position[0].cpu(), kin.joint_limits_position[1].cpu()
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo) * scale
    q = (mid + half * (2 * torch.rand(num, kin.num_dof, generator=g) - 1)).to(kin.device)
    fk = Kinematics(KinematicsCfg(kin, None), compute_spheres=False)
    st = fk.compute_kinematics(q)
    return st.tool_poses.position[:, 0, 0].clone(), st.tool_poses.quaternion[:, 0, 0].clone()


def feasible_goals(kin, scene, num: int):
    """Goal poses = FK of COLLISION-FREE joint samples (the reference's ik_benchmark.py:92-101 protocol:
    ``ik_solver.sample_configs`` rejection-samples feasible configurations, then takes their tool
    poses), so that every goal has at least one collision-free solution.  Uses the product collision
    checker (HIP kernels), not the oracle."""
    from .collision_checking import RobotCollisionChecker
    from .kinematics import Kinematics, KinematicsCfg

    cfg = KinematicsCfg(kin, None)
    q = RobotCollisionChecker(cfg, scene).sample(num, mask_valid=True)
    if q.shape[0] < num:
        raise RuntimeError(f"rejection sampling found only {q.shape[0]} of {num} collision-free configurations")
    st = Kinematics(cfg, compute_spheres=False).compute_kinematics(q.contiguous())
    return st.tool_poses.position[:, 0, 0].clone(), st.tool_poses.quaternion[:, 0, 0].clone()


def c3_voxel_world(n: int = 128, voxel_size: float = 0.02) -> Dict:
    """C3 "UR10 + nvblox ESDF voxel world (128^3)": one fp16 ESDF grid (2.56 m cube at 0.02 m) around
    the robot, synthesised analytically from the union of a box and a sphere (the reference's
    voxel suites use the same kind of analytic fields, tests/_src/geom/sdf/test_voxel_collision.py)."""
    from .scene import voxel_grid_from_sdf

    def sdf(p):
        qb = np.abs(p - np.array([0.5, 0.0, 0.35])) - np.array([0.2, 0.4, 0.35])
        box = np.linalg.norm(np.maximum(qb, 0), axis=-1) + np.minimum(qb.max(-1), 0)
        sp = np.linalg.norm(p - np.array([-0.35, 0.45, 0.6]), axis=-1) - 0.25
        return np.minimum(box, sp)

    return voxel_grid_from_sdf(sdf, (n, n, n), voxel_size, pose7=(0.0, 0.0, 0.6, 1, 0, 0, 0), max_distance=10.0)


def c5_mixed_worlds(num_envs: int, voxels: bool = True, grid: int = 64, seed: int = 5, rotated: bool = False) -> Dict:
    """C5 "mixed scene": every planning problem has its OWN world -- a table plus 1-3 random cuboids
    and (``voxels``) one fp16 ESDF grid (``grid``^3 at 0.04 m) holding a sphere obstacle.  ``rotated``: the random
    cuboids are turned about two axes (parity tests: in a rotated obstacle frame the swept cost's stationary-sphere
    branch depends on rounding); the benchmark's worlds are axis aligned."""
    from .scene import cuboid_scene_arrays, voxel_grid_from_sdf

    rng = np.random.default_rng(seed)
    envs, grids = [], []
    for e in range(num_envs):
        obs = [{"dims": [2.2, 2.2, 0.2], "pose": [0.0, 0.0, -0.1, 1, 0, 0, 0]}]
        for k in range(1 + e % 3):
            p = rng.uniform([-0.6, -0.6, 0.2], [0.6, 0.6, 0.9])
            quat = [1, 0, 0, 0]
            if rotated:
                a, b = 0.3 + 0.4 * k + 0.2 * e, 0.2 + 0.1 * k
                qz, qx = np.array([np.cos(a / 2), 0, 0, np.sin(a / 2)]), np.array([np.cos(b / 2), np.sin(b / 2), 0, 0])
                quat = [qz[0] * qx[0] - qz[3] * 0, qz[0] * qx[1], qz[3] * qx[1], qz[3] * qx[0]]  # qz (x) qx, wxyz
            obs.append({"dims": list(rng.uniform(0.1, 0.35, size=3)), "pose": [*p, *quat]})
        envs.append(obs)
        c = rng.uniform([-0.5, -0.5, 0.3], [0.5, 0.5, 0.8])
        grids.append(voxel_grid_from_sdf(lambda p, c=c: np.linalg.norm(p - c, axis=-1) - 0.15, (grid,) * 3, 0.04,
                                         pose7=(0.0, 0.0, 0.6, 1, 0, 0, 0), max_distance=10.0))
    arrays = cuboid_scene_arrays(envs)
    if voxels:
        for k in ("voxel_params", "voxel_inv_pose", "voxel_enable", "voxel_count", "voxel_features"):
            arrays[k] = np.concatenate([g[k] for g in grids], axis=0)
        arrays["voxel_max_distance"] = grids[0]["voxel_max_distance"]
    return arrays
