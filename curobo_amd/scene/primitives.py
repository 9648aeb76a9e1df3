"""Non-cuboid obstacles for the sphere-world collision kernels: analytic signed distance fields of
the reference's primitive types and a triangle-mesh signed distance, baked into the fp16 ESDF
voxel grids that ``csrc/scene_collision.hip`` already consumes.

The reference routes every non-cuboid primitive through a mesh (``geom/types.py:1104-1124``:
``Sphere`` / ``Capsule`` / ``Cylinder`` -> ``get_mesh`` -> Warp's BVH ``wp.mesh_query_point``,
``geom/data/data_mesh.py:555-700``).  Warp is a third-party dependency outside the reference tree
and its mesh SDF has no numeric test upstream (parity unpinned), so this is a
host-side route with the same semantics (negative inside, world-frame pose ``[x y z qw qx qy qz]``):

* primitives: closed-form SDFs in the obstacle frame (``geom/types.py:290-450`` field names);
* meshes: exact point-triangle distance (vectorised NumPy) with the sign from the generalised
  winding number (closed, consistently oriented meshes; robust at edges and vertices);
* ``bake_esdf`` samples any of them on a voxel grid (``voxel_grid_from_sdf``), the layout of the
  reference's ``VoxelGrid`` / ``VoxelData`` (``geom/data/data_voxel.py:42-95``).

Baking happens once per scene update on the host; the per-step path only sees the voxel grid.
"""

from __future__ import annotations

from typing import Callable, Dict, Sequence, Tuple

import numpy as np

from .data import inverse_pose7, voxel_grid_from_sdf


def _to_local(pose7: Sequence[float], pts: np.ndarray) -> np.ndarray:
    """world points -> obstacle frame (pose7 = [x y z qw qx qy qz] of the obstacle in the world)"""
    inv = np.asarray(inverse_pose7(pose7), np.float64)
    w, x, y, z = inv[3:7]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return np.asarray(pts, np.float64) @ R.T + inv[:3]


SDF = Callable[[np.ndarray], np.ndarray]


def sphere_sdf(radius: float, pose7: Sequence[float] = (0, 0, 0, 1, 0, 0, 0)) -> SDF:
    c = np.asarray(pose7[:3], np.float64)
    return lambda p: np.linalg.norm(np.asarray(p, np.float64) - c, axis=-1) - radius


def cuboid_sdf(dims: Sequence[float], pose7: Sequence[float] = (0, 0, 0, 1, 0, 0, 0)) -> SDF:
    h = 0.5 * np.asarray(dims, np.float64)

    def f(p):
        q = np.abs(_to_local(pose7, p)) - h
        return np.linalg.norm(np.maximum(q, 0.0), axis=-1) + np.minimum(q.max(-1), 0.0)
    return f


def capsule_sdf(radius: float, base: Sequence[float], tip: Sequence[float], pose7: Sequence[float] = (0, 0, 0, 1, 0, 0, 0)) -> SDF:
    """segment base-tip (obstacle frame) swept by a sphere (reference Capsule: radius, base, tip)"""
    a, b = np.asarray(base, np.float64), np.asarray(tip, np.float64)
    ab = b - a
    den = max(float(ab @ ab), 1e-30)

    def f(p):
        q = _to_local(pose7, p)
        t = np.clip(((q - a) @ ab) / den, 0.0, 1.0)
        return np.linalg.norm(q - a - t[..., None] * ab, axis=-1) - radius
    return f


def cylinder_sdf(radius: float, height: float, pose7: Sequence[float] = (0, 0, 0, 1, 0, 0, 0)) -> SDF:
    """axis = obstacle-frame z, centred at the origin (trimesh.creation.cylinder convention of the reference)"""
    def f(p):
        q = _to_local(pose7, p)
        d = np.stack([np.linalg.norm(q[..., :2], axis=-1) - radius, np.abs(q[..., 2]) - 0.5 * height], -1)
        return np.minimum(d.max(-1), 0.0) + np.linalg.norm(np.maximum(d, 0.0), axis=-1)
    return f


def union_sdf(*fields: SDF) -> SDF:
    return lambda p: np.min(np.stack([f(p) for f in fields], 0), axis=0)


# ---------------------------------------------------------------------------------------------- meshes
def _point_triangle_distance2(p: np.ndarray, a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """squared distance from points p[Q,3] to triangles (a,b,c)[F,3] -> [Q,F] (Ericson, Real-Time
    Collision Detection 5.1.5, region tests written with masks)"""
    ab, ac = (b - a)[None], (c - a)[None]
    ap = p[:, None, :] - a[None]
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p[:, None, :] - b[None]
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p[:, None, :] - c[None]
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        den = 1.0 / (va + vb + vc)
        v_in, w_in = vb * den, vc * den
        v_ab = d1 / (d1 - d3)
        w_ac = d2 / (d2 - d6)
        w_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
    closest = a[None] + ab * v_in[..., None] + ac * w_in[..., None]  # interior

    def put(mask, val):
        nonlocal closest
        closest = np.where(mask[..., None], val, closest)
    put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b[None] + w_bc[..., None] * (c - b)[None])
    put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a[None] + w_ac[..., None] * ac)
    put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a[None] + v_ab[..., None] * ab)
    put((d6 >= 0) & (d5 <= d6), np.broadcast_to(c[None], closest.shape))
    put((d3 >= 0) & (d4 <= d3), np.broadcast_to(b[None], closest.shape))
    put((d1 <= 0) & (d2 <= 0), np.broadcast_to(a[None], closest.shape))
    diff = p[:, None, :] - closest
    return (diff * diff).sum(-1)


def _winding_number(p: np.ndarray, a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """generalised winding number (sum of signed solid angles / 4 pi, van Oosterom-Strackee) -> [Q]"""
    A, B, C = a[None] - p[:, None], b[None] - p[:, None], c[None] - p[:, None]
    la, lb, lc = np.linalg.norm(A, axis=-1), np.linalg.norm(B, axis=-1), np.linalg.norm(C, axis=-1)
    num = (A * np.cross(B, C)).sum(-1)
    den = la * lb * lc + (A * B).sum(-1) * lc + (B * C).sum(-1) * la + (C * A).sum(-1) * lb
    return (2.0 * np.arctan2(num, den)).sum(-1) / (4.0 * np.pi)


def mesh_sdf(vertices: np.ndarray, faces: np.ndarray, pose7: Sequence[float] = (0, 0, 0, 1, 0, 0, 0), chunk: int = 2048) -> SDF:
    """signed distance to a closed, consistently oriented triangle mesh (vertices in the obstacle frame)"""
    v = np.asarray(vertices, np.float64)
    f = np.asarray(faces, np.int64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]

    def field(p):
        q = _to_local(pose7, np.asarray(p, np.float64).reshape(-1, 3))
        out = np.empty(q.shape[0])
        for i in range(0, q.shape[0], chunk):
            s = q[i:i + chunk]
            d = np.sqrt(_point_triangle_distance2(s, a, b, c).min(-1))
            inside = np.abs(_winding_number(s, a, b, c)) > 0.5
            out[i:i + chunk] = np.where(inside, -d, d)
        return out.reshape(np.asarray(p).shape[:-1])
    return field


def box_mesh(dims: Sequence[float]) -> Tuple[np.ndarray, np.ndarray]:
    """12-triangle mesh of an axis-aligned box centred at the origin, outward normals"""
    h = 0.5 * np.asarray(dims, np.float64)
    v = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float64) * h
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6],
                  [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.int64)
    return v, f


def bake_esdf(field: SDF, bounds_min: Sequence[float], bounds_max: Sequence[float], voxel_size: float,
              max_distance: float = 100.0) -> Dict[str, np.ndarray]:
    """axis-aligned fp16 ESDF grid covering [bounds_min, bounds_max] (world frame), ready for
    ``SceneData.from_arrays({**cuboid_arrays, **grid})``"""
    lo, hi = np.asarray(bounds_min, np.float64), np.asarray(bounds_max, np.float64)
    shape = np.maximum(np.ceil((hi - lo) / voxel_size).astype(int), 2)
    centre = 0.5 * (lo + hi)
    return voxel_grid_from_sdf(lambda p: np.clip(field(p), -max_distance, max_distance), tuple(int(s) for s in shape),
                               voxel_size, pose7=(*centre, 1, 0, 0, 0), max_distance=max_distance)


def bake_mesh_esdf_device(vertices, faces, mesh_pose7: Sequence[float], bounds_min: Sequence[float], bounds_max: Sequence[float],
                          voxel_size: float, device, max_distance: float = 100.0) -> Dict[str, object]:
    """A closed triangle mesh (``vertices`` [V, 3] in the mesh frame, ``faces`` [F, 3], world pose ``mesh_pose7``) baked
    ON THE DEVICE into an axis-aligned fp16 ESDF grid covering [bounds_min, bounds_max] (world frame): the device
    counterpart of ``bake_esdf(mesh_sdf(...))`` (same algorithm: exact point-triangle distance + winding-number sign,
    csrc/mesh_bake.hip).  Returns the voxel arrays for ``SceneData.from_arrays`` (``voxel_features`` is a device tensor)."""
    import torch

    from ..backends.collision import mesh_esdf_bake

    lo, hi = np.asarray(bounds_min, np.float64), np.asarray(bounds_max, np.float64)
    shape = tuple(int(v) for v in np.maximum(np.ceil((hi - lo) / voxel_size).astype(int), 2))
    centre = 0.5 * (lo + hi)
    # grid frame (axis aligned, origin at the grid centre) -> world -> mesh frame
    inv = np.asarray(inverse_pose7(mesh_pose7), np.float64)
    w, x, y, z = inv[3:7]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    t = R @ centre + inv[:3]
    g2m = np.concatenate([R, t[:, None]], axis=1).reshape(-1)
    dev = torch.device(device)
    out = torch.empty(shape[0] * shape[1] * shape[2], dtype=torch.float16, device=dev)
    v = torch.as_tensor(np.ascontiguousarray(vertices, dtype=np.float32), device=dev)
    f = torch.as_tensor(np.ascontiguousarray(faces, dtype=np.int32), device=dev)
    with torch.cuda.device(dev):
        mesh_esdf_bake(out, v, f, shape, voxel_size, g2m, max_distance)
    inv_pose = np.zeros((1, 1, 8), np.float32)
    inv_pose[0, 0, :7] = inverse_pose7((*centre, 1, 0, 0, 0))
    return {"voxel_params": np.array([[[shape[0], shape[1], shape[2], voxel_size]]], np.float32), "voxel_inv_pose": inv_pose,
            "voxel_enable": np.ones((1, 1), np.uint8), "voxel_count": np.ones((1,), np.int32),
            "voxel_features": out.view(1, 1, -1), "voxel_max_distance": float(max_distance)}
