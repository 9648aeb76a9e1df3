"""World obstacle stores in the layout the collision kernels read.

Mirrors the tensor layouts of the reference's ``CuboidData`` (``curobo/_src/geom/data/data_cuboid.py``
``:67-108``: ``dims [E,n,4]`` full extents, ``inv_pose [E,n,8]`` = x y z qw qx qy qz pad,
``enable u8 [E,n]``, ``count i32 [E]``) and ``VoxelData`` (``geom/data/data_voxel.py:42-95``:
``params [E,n,4]`` = nx ny nz voxel_size, ``features fp16 [E,n,nvox]``).  Arrays are built with
numpy (shared by the oracle tests) and uploaded once; the hot path only reads them.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


def _quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def _quat_rotate(q, v):
    w, x, y, z = q
    qv = np.array([x, y, z])
    t = 2.0 * np.cross(qv, v)
    return v + w * t + np.cross(qv, t)


def inverse_pose7(pose7: Sequence[float]) -> np.ndarray:
    """[x y z qw qx qy qz] -> inverse pose in the same layout (world->obstacle frame)."""
    p = np.asarray(pose7[:3], dtype=np.float64)
    q = np.asarray(pose7[3:7], dtype=np.float64)
    q = q / np.linalg.norm(q)
    qi = np.array([q[0], -q[1], -q[2], -q[3]])
    pi = -_quat_rotate(qi, p)
    return np.concatenate([pi, qi])


def cuboid_scene_arrays(envs: List[List[Dict]], max_n: Optional[int] = None) -> Dict[str, np.ndarray]:
    """``envs[e]`` = list of ``{"dims": [x,y,z], "pose": [x,y,z,qw,qx,qy,qz], "enable": bool}`` cuboids and / or analytic
    primitives (reference ``Sphere`` / ``Capsule`` / ``Cylinder``, geom/types.py:290-450; evaluated in closed form on
    the device instead of through a mesh): ``{"type": "sphere", "radius": r, "pose": ...}``, ``{"type": "capsule",
    "radius": r, "base": [0,0,z0], "tip": [0,0,z1], "pose": ...}`` (segment on the local z axis, as the reference
    requires for its capsule mesh) and ``{"type": "cylinder", "radius": r, "height": h, "pose": ...}`` (axis = local z,
    centred).  They live in the cuboid store with a tag in ``dims[..., 3]`` (see include/curobo_hip.h)."""
    E = len(envs)
    n = max_n or max(1, max(len(e) for e in envs))
    if any(len(e) > n for e in envs):
        raise ValueError(f"Cannot load {max(len(e) for e in envs)} cuboids, max cache size is {n}")
    dims = np.zeros((E, n, 4), np.float32)
    inv_pose = np.zeros((E, n, 8), np.float32)
    inv_pose[..., 3] = 1.0
    enable = np.zeros((E, n), np.uint8)
    count = np.zeros((E,), np.int32)
    for e, obs in enumerate(envs):
        count[e] = len(obs)
        for i, o in enumerate(obs):
            kind, pose = o.get("type", "cuboid"), list(o["pose"])
            if kind == "cuboid":
                dims[e, i, :3] = o["dims"]
            elif kind == "sphere":
                dims[e, i] = [o["radius"], 0.0, 0.0, 1.0]
            elif kind == "cylinder":
                dims[e, i] = [o["radius"], 0.5 * o["height"], 0.0, 3.0]
            elif kind == "capsule":
                base, tip = np.asarray(o.get("base", [0, 0, 0]), np.float64), np.asarray(o.get("tip", [0, 0, 0]), np.float64)
                if abs(base[0]) + abs(base[1]) + abs(tip[0]) + abs(tip[1]) > 0 or tip[2] < base[2]:
                    raise ValueError("capsule base / tip must lie on the local z axis with tip above base (as in the reference)")
                dims[e, i] = [o["radius"], 0.5 * (tip[2] - base[2]), 0.0, 2.0]
                # the stored frame is centred on the segment: shift the pose along the obstacle's own z axis
                q = np.asarray(pose[3:7], np.float64)
                pose[:3] = list(np.asarray(pose[:3], np.float64) + _quat_rotate(q / np.linalg.norm(q), np.array([0, 0, 0.5 * (base[2] + tip[2])])))
            else:
                raise ValueError(f"unknown obstacle type {kind!r}")
            inv_pose[e, i, :7] = inverse_pose7(pose)
            enable[e, i] = 1 if o.get("enable", True) else 0
    names = [[o.get("name") for o in obs] + [None] * (n - len(obs)) for obs in envs]  # (a list, not an array: stays on the host)
    return {"cuboid_dims": dims, "cuboid_inv_pose": inv_pose, "cuboid_enable": enable, "cuboid_count": count, "cuboid_names": names}


def voxel_grid_from_sdf(sdf_fn: Callable[[np.ndarray], np.ndarray], grid_shape: Sequence[int],
                        voxel_size: float, pose7: Sequence[float] = (0, 0, 0, 1, 0, 0, 0),
                        max_distance: float = 10000.0) -> Dict[str, np.ndarray]:
    """One fp16 ESDF grid sampled at voxel centres (align-corners convention of the reference:
    voxel (i,j,k) centre = ((i,j,k) + 0.5 - n/2) * voxel_size in the grid frame)."""
    nx, ny, nz = [int(v) for v in grid_shape]
    ax = [(np.arange(n) + 0.5 - n / 2.0) * voxel_size for n in (nx, ny, nz)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    pts_local = np.stack([X, Y, Z], -1).reshape(-1, 3)
    # the field is defined in world coordinates; grid frame -> world
    q = np.asarray(pose7[3:7], dtype=np.float64)
    q = q / np.linalg.norm(q)
    pts_world = np.stack([_quat_rotate(q, p) for p in pts_local]) + np.asarray(pose7[:3]) \
        if not np.allclose(q, [1, 0, 0, 0]) else pts_local + np.asarray(pose7[:3])
    vals = sdf_fn(pts_world).astype(np.float16).reshape(1, 1, -1)
    params = np.array([[[nx, ny, nz, voxel_size]]], np.float32)
    inv_pose = np.zeros((1, 1, 8), np.float32)
    inv_pose[0, 0, :7] = inverse_pose7(pose7)
    return {
        "voxel_params": params, "voxel_inv_pose": inv_pose, "voxel_enable": np.ones((1, 1), np.uint8),
        "voxel_count": np.ones((1,), np.int32), "voxel_features": vals,
        "voxel_max_distance": float(max_distance),
    }


@dataclass
class SceneData:
    """Device-resident scene; ``.struct`` is the ``curobo_hip_scene`` passed to the kernels."""

    tensors: Dict[str, "object"]
    struct: "object"
    arrays: Dict[str, np.ndarray]
    #: mesh obstacles (``scene.mesh.MeshStore``): queried through their BVHs by a launch of its own after the scene launch
    #: (``backends.collision.sphere_obstacle_collision`` runs it when ``struct.mesh_set`` is set)
    meshes: Optional["object"] = None
    #: the description the stores were built from (``SceneCfg``, or a list with one per environment) when there was one:
    #: what ``AttachmentManager.attach_from_scene`` looks obstacles up in (reference ``SceneCollision.scene_model``)
    scene_model: Optional["object"] = None

    @property
    def num_envs(self) -> int:
        """number of scene environments (leading dimension of the obstacle stores)"""
        for k in ("cuboid_dims", "voxel_params"):
            if k in self.arrays:
                return int(self.arrays[k].shape[0])
        if self.meshes is not None:  # a mesh-only scene: the mesh store carries the environment count
            return int(self.meshes.num_envs)
        return 1

    # ------------------------------------------------------------------ obstacles by name (reference SceneData / SceneCollision:
    # geom/data/data_scene.py:273-417, data_cuboid.py:259-428).  Every change is an in-place write into the tensors the kernels
    # (and captured graphs) read; the host copies in ``arrays`` follow.
    def _names(self, kind: str, env_idx: int) -> List[Optional[str]]:
        if kind == "mesh":
            return list(self.meshes.names[env_idx]) if self.meshes is not None else []
        names = self.arrays.get(f"{kind}_names")
        return names[env_idx] if names is not None else []

    def _count(self, kind: str, env_idx: int) -> int:
        c = self.arrays.get(f"{kind}_count")
        return int(c[env_idx]) if c is not None else 0

    def find_obstacle(self, name: str, env_idx: int = 0):
        """(kind, slot) of a named obstacle: cuboid store (cuboids and analytic primitives), meshes, voxel grids, in the
        reference's order of search; ``ValueError`` when no store of this environment holds the name"""
        for kind in ("cuboid", "mesh", "voxel"):
            names = self._names(kind, env_idx)
            if name in names:
                return kind, names.index(name)
        raise ValueError(f"Obstacle '{name}' not found in environment {env_idx}")

    def get_obstacle_names(self, env_idx: int = 0) -> List[str]:
        out = list(self._names("cuboid", env_idx)[: self._count("cuboid", env_idx)])
        out += [n for n in self._names("mesh", env_idx) if n is not None]
        out += list(self._names("voxel", env_idx)[: self._count("voxel", env_idx)])
        return out

    def check_obstacle_exists(self, name: str, env_idx: int = 0) -> bool:
        return name in self.get_obstacle_names(env_idx)

    def _write(self, key: str, index, value) -> None:
        import torch

        t = self.tensors[key]
        t[index] = torch.as_tensor(value, dtype=t.dtype, device=t.device)
        if isinstance(self.arrays.get(key), np.ndarray):
            self.arrays[key][index] = np.asarray(value.detach().cpu().numpy() if hasattr(value, "detach") else value, dtype=self.arrays[key].dtype)

    def enable_obstacle(self, name: str, enable: bool = True, env_idx: int = 0) -> None:
        kind, i = self.find_obstacle(name, env_idx)
        if kind == "mesh":
            self.meshes.set_enabled(name, enable, env_idx)
        else:
            self._write(f"{kind}_enable", (env_idx, i), int(bool(enable)))

    @staticmethod
    def _pose7(pose) -> List[float]:
        if hasattr(pose, "position") and hasattr(pose, "quaternion"):  # a ``Pose``
            return [float(v) for v in pose.position.reshape(-1)[:3].tolist() + pose.quaternion.reshape(-1)[:4].tolist()]
        return [float(v) for v in pose]

    def update_obstacle_pose(self, name: str, w_obj_pose, env_idx: int = 0) -> None:
        """the obstacle's new pose in the world frame (a ``Pose`` or [x y z qw qx qy qz]); the stores hold its inverse"""
        kind, i = self.find_obstacle(name, env_idx)
        pose = self._pose7(w_obj_pose)
        if kind == "mesh":
            self.meshes.update_pose(name, pose, env_idx)
            return
        if kind == "cuboid" and float(self.arrays["cuboid_dims"][env_idx, i, 3]) == 2.0:
            raise ValueError(f"'{name}' is a capsule: its stored frame is centred on its segment; re-load the scene to move it")
        self._write(f"{kind}_inv_pose", (env_idx, i, slice(0, 7)), inverse_pose7(pose).astype(np.float32))

    def update_obstacle_dims(self, name: str, dims, env_idx: int = 0) -> None:
        """new full extents [x, y, z] of a cuboid (reference CuboidData.update_dims)"""
        kind, i = self.find_obstacle(name, env_idx)
        if kind != "cuboid" or float(self.arrays["cuboid_dims"][env_idx, i, 3]) != 0.0:
            raise ValueError(f"'{name}' is not a cuboid")
        self._write("cuboid_dims", (env_idx, i, slice(0, 3)), np.asarray(dims, np.float32))

    def add_obstacle(self, obstacle, env_idx: int = 0) -> int:
        """a cuboid / sphere / capsule / cylinder (``scene.types`` object or its dictionary with ``name``) into the next free
        slot of the cuboid store of one environment; the store's capacity is what the scene was built with (``max_n`` of
        ``cuboid_scene_arrays``, the reference's ``cache``): a full store or a name already there raises as the reference's does"""
        if hasattr(obstacle, "get_mesh_data") or hasattr(obstacle, "get_grid_shape"):
            raise ValueError("add_obstacle takes cuboids and analytic primitives; meshes and voxel grids are loaded with the scene")
        if not isinstance(obstacle, dict):
            from .types import SceneCfg

            cfg = SceneCfg()
            cfg.add_obstacle(obstacle)
            from .config import _one_env

            obstacle = _one_env(cfg.to_config())[0]
        if "cuboid_dims" not in self.arrays:
            raise ValueError("the scene has no cuboid store")
        n, cap = self._count("cuboid", env_idx), int(self.arrays["cuboid_dims"].shape[1])
        if n >= cap:
            raise RuntimeError(f"Cannot add cuboid, cache is full ({cap} cuboids)")
        if obstacle.get("name") in self._names("cuboid", env_idx)[:n]:
            raise RuntimeError(f"Cuboid already exists with name: {obstacle.get('name')}")
        one = cuboid_scene_arrays([[obstacle]], max_n=1)
        self._write("cuboid_dims", (env_idx, n), one["cuboid_dims"][0, 0])
        self._write("cuboid_inv_pose", (env_idx, n), one["cuboid_inv_pose"][0, 0])
        self._write("cuboid_enable", (env_idx, n), one["cuboid_enable"][0, 0])
        self._write("cuboid_count", (env_idx,), n + 1)
        self.arrays.setdefault("cuboid_names", [[None] * cap for _ in range(self.num_envs)])[env_idx][n] = obstacle.get("name")
        return n

    def clear(self, env_idx: Optional[int] = None) -> None:
        """Remove every cuboid / voxel-grid / mesh obstacle of one environment (``None``: of all), in place (the kernels and
        captured graphs read the same enable / count tensors): an empty world of the same capacity.  As the reference's
        stores do it (geom/data/data_cuboid.py:411-424, data_voxel.py:624-639, data_mesh.py:472-486): enable flags AND
        counts to zero, the names forgotten -- so that a cleared obstacle is not found by name any more, cannot be
        switched back on, and the same world can be added again into the freed slots."""
        sel = slice(None) if env_idx is None else env_idx
        envs = range(self.num_envs) if env_idx is None else [env_idx]
        for kind in ("cuboid", "voxel"):
            for key in (f"{kind}_enable", f"{kind}_count"):
                if self.tensors.get(key) is not None:
                    self.tensors[key][sel] = 0
                if isinstance(self.arrays.get(key), np.ndarray):
                    self.arrays[key][sel] = 0
            names = self.arrays.get(f"{kind}_names")
            if names is not None:
                for e in envs:
                    names[e] = [None] * len(names[e])
        if self.meshes is not None and getattr(self.meshes, "enable", None) is not None:
            self.meshes.enable[sel] = 0
            self.meshes.count[sel] = 0
            for e in envs:
                self.meshes.names[e] = [None] * len(self.meshes.names[e])

    @staticmethod
    def from_arrays(arrays: Optional[Dict[str, np.ndarray]], device, coarse_culling: bool = True, meshes=None) -> "SceneData":
        """``coarse_culling``: also build the min-pooled ESDF the voxel kernels use to skip spheres that are far from
        every surface (``curobo_hip_scene.voxel_coarse_min``; results are identical with and without it).  ``meshes``:
        per environment a list of mesh obstacles (see ``scene.mesh.MeshStore``), or a ready ``MeshStore``."""
        arrays = arrays if arrays is not None else {}
        import torch

        from ..backends.collision import build_voxel_coarse_min, make_scene

        t = {}
        for k, v in arrays.items():
            if isinstance(v, np.ndarray):
                t[k] = torch.as_tensor(v).to(device).contiguous()
            elif torch.is_tensor(v):  # e.g. an ESDF grid baked on the device (bake_mesh_esdf_device)
                t[k] = v.to(device).contiguous()
        coarse, block, dilate = None, 0, 0
        if coarse_culling and t.get("voxel_features") is not None and t["voxel_features"].numel() > 0:
            block, dilate = 4, 3  # culls spheres whose sweep stays within 2 voxels of the centre's voxel
            import os
            if os.environ.get("CUROBO_VOXEL_COARSE"):  # development knob: "block,dilate" (a malformed value is ignored)
                try:
                    b_, d_ = (int(v) for v in os.environ["CUROBO_VOXEL_COARSE"].split(","))
                    if b_ >= 1 and d_ >= 0:
                        block, dilate = b_, d_
                except ValueError:
                    pass
            coarse = t["voxel_coarse_min"] = build_voxel_coarse_min(t["voxel_features"], arrays["voxel_params"], block, dilate)
        struct = make_scene(
            t.get("cuboid_dims"), t.get("cuboid_inv_pose"), t.get("cuboid_enable"), t.get("cuboid_count"),
            t.get("voxel_params"), t.get("voxel_inv_pose"), t.get("voxel_enable"), t.get("voxel_count"),
            t.get("voxel_features"), float(arrays.get("voxel_max_distance", 10000.0)),
            voxel_coarse_min=coarse, voxel_coarse_block=block, voxel_coarse_dilate=dilate,
        )
        store = None
        if meshes is not None:
            from .mesh import MeshStore

            store = meshes if isinstance(meshes, MeshStore) else MeshStore(meshes, device)
            # the mesh launch indexes its per-environment tables (count, mesh_id, inv_pose, enable) with the SAME
            # env_query_idx as the cuboid / voxel stores: a store with fewer environments would be read out of bounds
            n_arr = [int(arrays[k].shape[0]) for k in ("cuboid_dims", "voxel_params") if k in arrays]
            if n_arr and any(n != int(store.num_envs) for n in n_arr):
                raise ValueError(f"meshes describe {store.num_envs} environment(s) but the obstacle arrays {n_arr[0]}: "
                                 "every obstacle store of a scene must cover the same environments")
            struct.mesh_set = store.struct  # (a Python attribute of the ctypes struct: the launch wrappers look for it)
        return SceneData(tensors=t, struct=struct, arrays=arrays, meshes=store)


def warn_if_reference_mesh_gradient(scene: Optional["SceneData"], who: str) -> None:
    """a solver over a scene whose mesh store hands out the reference's mesh gradient as written (``MeshStore`` mode 0): say once
    what that does to an optimiser -- the scenes ``scene_from_config`` builds use the consistent vector"""
    meshes = getattr(scene, "meshes", None) if scene is not None else None
    if meshes is not None and getattr(meshes, "gradient_mode", 1) == 0 and not getattr(warn_if_reference_mesh_gradient, "_said", False):
        import warnings

        warn_if_reference_mesh_gradient._said = True
        warnings.warn(f"{who}: the scene's mesh store returns the reference's mesh gradient as written (gradient_mode 0): for a sphere "
                      "outside a mesh it points away from what the cuboid and voxel queries return, and optimisers are held inside meshes "
                      "by it.  Build the store with MeshStore(..., gradient_mode=MeshStore.CONSISTENT_GRADIENT) (what scene_from_config "
                      "does) unless reproducing data_mesh.py:693-697 is the point.", stacklevel=3)


def validate_env_query_idx(env_query_idx, scene: Optional["SceneData"], kin_num_envs: int = 1) -> None:
    """Reject environment indices the kernels would read out of bounds: every entry must address one of
    the scene's environments and, when the robot carries per-environment collision spheres
    (``KinematicsParams.num_envs > 1``: the same index selects the sphere table in the FK kernels), one of
    those as well.  One host read-back, at update time (never inside a captured launch sequence)."""
    import torch

    if env_query_idx is None or env_query_idx.numel() == 0:
        return
    lo, hi = int(torch.min(env_query_idx)), int(torch.max(env_query_idx))
    limits = []
    if scene is not None:
        limits.append(("scene environments", scene.num_envs))
    if kin_num_envs > 1:
        limits.append(("robot sphere environments", int(kin_num_envs)))
    for what, n in limits:
        if lo < 0 or hi >= n:
            raise ValueError(f"env_query_idx spans [{lo}, {hi}] but there are {n} {what}")
