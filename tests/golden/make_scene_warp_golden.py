"""Golden vectors for sphere-obstacle collision, produced by the REFERENCE's own Warp kernels executed on the CPU.

    PYTHONPATH=/root/reference python tests/golden/make_scene_warp_golden.py

NVIDIA Warp is not installed here, but a Warp kernel is a typed Python function: with ``tests/golden/warp_emulator`` on
the path as ``warp`` (vector / quaternion / transform algebra, C integer division, atomics, overloads by struct type:
see its docstring), the reference's unmodified sources are imported and run thread by thread, in fp32:

    curobo/_src/geom/collision/wp_collision_kernel.py        sphere_obstacle_collision_kernel
    curobo/_src/geom/collision/wp_sweep_collision_kernel.py  swept_sphere_obstacle_collision_kernel (SWEEP_STEPS = 3)
    curobo/_src/geom/collision/wp_speed_metric.py            apply_speed_metric
    curobo/_src/geom/collision/wp_collision_common.py        activation, accumulation
    curobo/_src/geom/data/data_cuboid.py, data_voxel.py      obstacle accessors, cuboid SDF, fp16 ESDF trilinear lookup
    curobo/_src/geom/data/helper_pose.py                     inverse-pose loads

The launches follow geom/collision/wp_autograd.py: outputs zeroed once, one launch per obstacle type (cuboids, then
voxel grids) with dim = spheres x max obstacles of that type, accumulating into the same buffers; the speed metric runs
once afterwards.  Threads run in index order (a legal schedule; the reference's float atomics make its own order
launch dependent), i.e. every sphere sums its obstacles in index order -- the canonical order of the oracle and of the
HIP kernels.  Inputs and outputs go to tests/golden/scene_warp_golden.npz, in the array layout of
``curobo_amd.scene`` / ``include/curobo_hip.h`` (which mirrors CuboidDataWarp / VoxelDataWarp).
"""
import importlib.abc
import importlib.machinery
import os
import sys
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "warp_emulator"))


class _StubMissing(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """third-party modules the reference's geometry types import at module level and these kernels never touch"""

    ROOTS = {"trimesh"}

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__, m.__name__, m.__spec__, m.__loader__ = [], spec.name, spec, self
        return m

    def exec_module(self, module):
        pass


sys.meta_path.append(_StubMissing())

import warp as wp  # noqa: E402  (the emulator)

from curobo._src.geom.collision.wp_collision_kernel import sphere_obstacle_collision_kernel  # noqa: E402
from curobo._src.geom.collision.wp_speed_metric import apply_speed_metric  # noqa: E402
from curobo._src.geom.collision.wp_sweep_collision_kernel import swept_sphere_obstacle_collision_kernel  # noqa: E402
from curobo._src.geom.data.data_cuboid import CuboidDataWarp  # noqa: E402
from curobo._src.geom.data.data_voxel import VoxelDataWarp  # noqa: E402

assert "emulator" in (wp.__doc__ or "") or "stand-in" in (wp.__doc__ or ""), "the real warp is on the path: not this script's case"


# ---------------------------------------------------------------- scene arrays (layout of curobo_amd.scene)
def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def quat_rot(q, v):
    qv = np.array([0.0, *v])
    qc = q * np.array([1, -1, -1, -1])
    return quat_mul(quat_mul(q, qv), qc)[1:]


def inverse_pose7(pose):
    p, q = np.asarray(pose[:3], np.float64), np.asarray(pose[3:7], np.float64)
    q = q / np.linalg.norm(q)
    qi = q * np.array([1, -1, -1, -1])
    return np.concatenate([-quat_rot(qi, p), qi]).astype(np.float32)


def cuboid_arrays(envs, max_n):
    E = len(envs)
    dims, inv = np.zeros((E, max_n, 4), np.float32), np.zeros((E, max_n, 8), np.float32)
    inv[..., 3] = 1.0
    en, cnt = np.zeros((E, max_n), np.uint8), np.zeros((E,), np.int32)
    for e, obs in enumerate(envs):
        cnt[e] = len(obs)
        for i, o in enumerate(obs):
            dims[e, i, :3] = o["dims"]
            inv[e, i, :7] = inverse_pose7(o["pose"])
            en[e, i] = 1 if o.get("enable", True) else 0
    return {"cuboid_dims": dims, "cuboid_inv_pose": inv, "cuboid_enable": en, "cuboid_count": cnt}


def voxel_arrays(envs, max_n, shape, voxel_size, max_dist):
    """envs[e] = list of {"pose", "sdf": f(points [n, 3]) -> [n], "enable"}; all grids share `shape` (the reference's
    n_voxels_per_layer is one number per store); voxel centres at (i - (n - 1) / 2) * voxel_size in the grid frame"""
    E, nv = len(envs), int(np.prod(shape))
    prm, inv = np.zeros((E, max_n, 4), np.float32), np.zeros((E, max_n, 8), np.float32)
    inv[..., 3] = 1.0
    prm[..., :3], prm[..., 3] = shape, voxel_size
    en, cnt = np.zeros((E, max_n), np.uint8), np.zeros((E,), np.int32)
    feat = np.full((E, max_n, nv), max_dist, np.float16)
    ax = [(np.arange(n, dtype=np.float64) - (n - 1) / 2.0) * voxel_size for n in shape]
    pts = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)
    for e, obs in enumerate(envs):
        cnt[e] = len(obs)
        for i, o in enumerate(obs):
            inv[e, i, :7] = inverse_pose7(o["pose"])
            en[e, i] = 1 if o.get("enable", True) else 0
            feat[e, i] = np.minimum(o["sdf"](pts), max_dist).astype(np.float16)
    return {"voxel_params": prm, "voxel_inv_pose": inv, "voxel_enable": en, "voxel_count": cnt, "voxel_features": feat,
            "voxel_max_distance": np.float32(max_dist)}


def box_sdf(half, centre=(0, 0, 0)):
    half, centre = np.asarray(half, np.float64), np.asarray(centre, np.float64)

    def f(p):
        q = np.abs(p - centre) - half
        return np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(-1), 0)

    return f


def ball_sdf(r, centre):
    centre = np.asarray(centre, np.float64)
    return lambda p: np.linalg.norm(p - centre, axis=-1) - r


def union(*fs):
    return lambda p: np.minimum.reduce([f(p) for f in fs])


# ---------------------------------------------------------------- the reference's structs and launches
def cuboid_struct(a):
    E, n = a["cuboid_dims"].shape[:2]
    return CuboidDataWarp(dims=wp.array(a["cuboid_dims"].reshape(E * n, 4), dtype=wp.float32),
                          inv_pose=wp.array(a["cuboid_inv_pose"].reshape(E * n, 8), dtype=wp.float32),
                          enable=wp.array(a["cuboid_enable"].reshape(-1), dtype=wp.uint8),
                          n_per_env=wp.array(a["cuboid_count"], dtype=wp.int32), max_n=wp.int32(n), num_envs=wp.int32(E))


def voxel_struct(a):
    E, n = a["voxel_params"].shape[:2]
    shape = a["voxel_params"][0, 0, :3].astype(int)
    bbox = np.zeros((E * n, 4), np.float32)
    bbox[:, :3] = a["voxel_params"].reshape(E * n, 4)[:, :3] * a["voxel_params"].reshape(E * n, 4)[:, 3:4]
    return VoxelDataWarp(params=wp.array(a["voxel_params"].reshape(E * n, 4), dtype=wp.float32), dims=wp.array(bbox, dtype=wp.float32),
                         inv_pose=wp.array(a["voxel_inv_pose"].reshape(E * n, 8), dtype=wp.float32),
                         enable=wp.array(a["voxel_enable"].reshape(-1), dtype=wp.uint8),
                         features=wp.array(a["voxel_features"].reshape(-1), dtype=wp.float16),
                         n_per_env=wp.array(a["voxel_count"], dtype=wp.int32), n_voxels_per_layer=wp.int32(int(np.prod(shape))),
                         max_n=wp.int32(n), num_envs=wp.int32(E), max_dist=wp.float32(a["voxel_max_distance"]))


def run(spheres, scene, weight, eta, env_idx, multi_env, swept, speed_dt):
    """the launch sequence of wp_autograd.py on zeroed buffers -> distance [B, H, S], gradient [B, H, S, 4]"""
    B, H, S, _ = spheres.shape
    n = B * H * S
    dist, grad = np.zeros(n, np.float32), np.zeros(n * 4, np.float32)
    sp = wp.array(spheres.reshape(n, 4).copy(), dtype=wp.vec4)
    kern = swept_sphere_obstacle_collision_kernel if swept else sphere_obstacle_collision_kernel
    sets = []
    if "cuboid_dims" in scene:
        sets.append((cuboid_struct(scene), scene["cuboid_dims"].shape[1]))
    if "voxel_params" in scene:
        sets.append((voxel_struct(scene), scene["voxel_params"].shape[1]))
    for obs, max_n in sets:
        wp.launch(kern, dim=n * max_n,
                  inputs=[obs, sp, wp.array(np.array([weight], np.float32)), wp.array(np.array([eta], np.float32)),
                          wp.array(np.asarray(env_idx, np.int32))],
                  outputs=[wp.array(dist), wp.array(grad), wp.int32(B), wp.int32(H), wp.int32(S), wp.int32(max_n),
                           wp.uint8(1 if multi_env else 0)])
    if speed_dt is not None:
        wp.launch(apply_speed_metric, dim=n,
                  inputs=[sp, wp.array(dist), wp.array(grad), wp.array(np.array([speed_dt], np.float32)), wp.int32(B), wp.int32(H),
                          wp.int32(S)])
    return dist.reshape(B, H, S), grad.reshape(B, H, S, 4)


# ---------------------------------------------------------------- cases
def trajectories(rng, B, H, S, centre, spread, step, radii):
    """smooth sphere trajectories through a region: start points, a per-trajectory drift, small per-sphere wobble"""
    start = centre + spread * rng.uniform(-1, 1, (B, 1, S, 3))
    drift = step * rng.uniform(-1, 1, (B, 1, 1, 3)) * np.arange(H).reshape(1, H, 1, 1)
    wob = 0.15 * step * rng.standard_normal((B, H, S, 3))
    sp = np.zeros((B, H, S, 4), np.float32)
    sp[..., :3] = start + drift + wob
    sp[..., 3] = rng.choice(radii, (B, 1, S))
    return sp


def main():
    rng = np.random.default_rng(20260924)
    rot = lambda ax, ang: [np.cos(ang / 2), *(np.sin(ang / 2) * np.asarray(ax) / np.linalg.norm(ax))]  # noqa: E731
    cub = cuboid_arrays([
        [{"dims": [0.6, 1.0, 0.05], "pose": [0.5, 0.0, 0.3, 1, 0, 0, 0]},                       # the reference's table fixture
         {"dims": [0.2, 0.3, 0.4], "pose": [0.35, 0.25, 0.55, *rot([0, 0, 1], 0.6)]},
         {"dims": [0.15, 0.15, 0.5], "pose": [0.45, -0.3, 0.6, *rot([1, 1, 0], 0.9)]},
         {"dims": [0.3, 0.3, 0.3], "pose": [0.2, 0.0, 0.8, *rot([0.3, -0.5, 0.8], 1.7)], "enable": False},
         {"dims": [0.1, 0.6, 0.1], "pose": [0.6, 0.1, 0.75, *rot([0, 1, 0], -0.4)]}],
        [{"dims": [0.4, 0.4, 0.4], "pose": [0.45, 0.05, 0.5, *rot([1, 0, 0], 0.3)]},
         {"dims": [0.05, 0.8, 0.6], "pose": [0.7, 0.0, 0.5, 1, 0, 0, 0]}],
    ], max_n=6)
    vox = voxel_arrays([
        [{"pose": [0.45, 0.0, 0.5, *rot([0, 0, 1], 0.35)],
          "sdf": union(box_sdf([0.12, 0.08, 0.1], [0.02, 0.0, -0.03]), ball_sdf(0.09, [-0.08, 0.1, 0.08]))}],
        [{"pose": [0.5, 0.1, 0.45, *rot([1, 2, 0], -0.5)], "sdf": box_sdf([0.05, 0.15, 0.12])},
         {"pose": [0.3, -0.2, 0.6, 1, 0, 0, 0], "sdf": ball_sdf(0.1, [0, 0, 0]), "enable": False}],
    ], max_n=2, shape=(26, 24, 22), voxel_size=0.02, max_dist=1000.0)
    B, H, S = 4, 6, 10
    radii = np.array([0.02, 0.035, 0.05, 0.07, -1.0], np.float32)  # a negative radius disables a sphere
    sp = trajectories(rng, B, H, S, np.array([0.45, 0.0, 0.5]), 0.3, 0.045, radii)
    sp[0, :, 0, :3] = sp[0, 0:1, 0, :3]  # one stationary sphere (half_dist = 0: the sweep loops do not run)
    sp[1, :, 1, :3] = [0.5, 0.0, 0.3]    # one sphere resting in the middle of the table (inside branch of the box SDF)
    sp[2, :, 2, :3] = [3.0, 3.0, 3.0]    # far outside every grid and box
    env_idx = np.array([0, 1, 1, 0], np.int32)
    out = {"spheres": sp, "env_query_idx": env_idx, **{k: v for k, v in cub.items()}, **{k: v for k, v in vox.items()}}
    both = {**cub, **vox}
    cases = [  # name, scene, weight, eta, multi_env, swept, speed_dt
        ("cuboid_static", cub, 1.0, 0.02, True, False, None),
        ("cuboid_static_eta0", cub, 2.5, 0.0, False, False, None),
        ("cuboid_swept", cub, 1.0, 0.02, True, True, None),
        ("voxel_static", vox, 1.0, 0.02, True, False, None),
        ("voxel_swept", vox, 3.0, 0.03, True, True, None),
        ("both_static", both, 1.0, 0.025, True, False, None),
        ("both_swept_speed", both, 5.0, 0.02, True, True, 0.05),
        ("both_static_speed", both, 1.0, 0.02, False, False, 0.02),
    ]
    meta = []
    for name, scene, w, eta, multi, swept, dt in cases:
        d, g = run(sp, scene, w, eta, env_idx, multi, swept, dt)
        out[f"{name}/distance"], out[f"{name}/gradient"] = d, g
        meta.append((name, w, eta, int(multi), int(swept), -1.0 if dt is None else dt,
                     int("cuboid_dims" in scene), int("voxel_params" in scene)))
        print(f"{name:20s} hits {int((d > 0).sum()):4d} / {d.size}  max {d.max():.4f}  |grad| max {np.abs(g).max():.4f}")
    out["case_names"] = np.array([m[0] for m in meta])
    out["case_params"] = np.array([m[1:] for m in meta], np.float64)  # weight, eta, multi_env, swept, speed_dt (-1 = off), cuboids, voxels
    path = os.path.join(HERE, "scene_warp_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
