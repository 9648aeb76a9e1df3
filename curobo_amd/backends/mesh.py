"""HIP backend for triangle-mesh obstacles: linear BVH build, closest-point / sign query, sphere-vs-mesh collision and the
ESDF bake through the BVH (``csrc/mesh_bvh.hip``).

The reference queries meshes through NVIDIA Warp (``curobo/_src/geom/data/data_mesh.py:555-700``: ``wp.mesh_query_point``
per query sphere; ``MeshData`` :60-120 holds the Warp mesh handles, poses, enable flags); Warp has no ROCm backend and is
not in the reference tree, so what these functions follow is the contract of that query as the reference uses it.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from .._lib import check, current_stream, load, ptr


class Mesh(C.Structure):
    """``curobo_hip_mesh``"""

    _fields_ = [("tri", C.c_void_p), ("node_box", C.c_void_p), ("n_tri", C.c_int32), ("n_leaves", C.c_int32),
                ("leaf_size", C.c_int32), ("_pad", C.c_int32)]


class MeshSet(C.Structure):
    """``curobo_hip_mesh_set``"""

    _fields_ = [("meshes", C.c_void_p), ("mesh_id", C.c_void_p), ("dims", C.c_void_p), ("inv_pose", C.c_void_p),
                ("enable", C.c_void_p), ("count", C.c_void_p), ("max_n", C.c_int32), ("gradient_mode", C.c_int32)]


@dataclass
class DeviceMesh:
    """one mesh on the device: sorted triangles + node boxes (kept alive here), the C struct, its bounding box"""

    tri: torch.Tensor
    node_box: torch.Tensor
    struct: Mesh
    bounds: np.ndarray  # [2, 3] lo / hi in the mesh frame
    n_tri: int

    @property
    def dims(self) -> np.ndarray:
        return (self.bounds[1] - self.bounds[0]).astype(np.float32)


def build_mesh_bvh(vertices, faces, device, leaf_size: int = 4) -> DeviceMesh:
    """vertices [V, 3] (mesh frame), faces [F, 3] -> the linear BVH on ``device``: Morton keys of the centroids (HIP), the
    sort (torch: plumbing), triangles in sorted order + node boxes (HIP, one launch per level)"""
    v = torch.as_tensor(np.ascontiguousarray(vertices, np.float32)).to(device).contiguous()
    f = torch.as_tensor(np.ascontiguousarray(faces, np.int32)).to(device).contiguous()
    if v.ndim != 2 or v.shape[1] != 3 or f.ndim != 2 or f.shape[1] != 3 or f.shape[0] == 0:
        raise ValueError(f"mesh needs vertices [V, 3] and faces [F, 3], got {tuple(v.shape)} and {tuple(f.shape)}")
    if int(f.min()) < 0 or int(f.max()) >= v.shape[0]:
        raise ValueError("face indices out of range")
    n = int(f.shape[0])
    used = v[f.reshape(-1).long()]
    bounds = np.stack([used.min(0).values.cpu().numpy(), used.max(0).values.cpu().numpy()]).astype(np.float32)
    codes = torch.empty(n, dtype=torch.int64, device=device)
    b = (C.c_float * 6)(*[float(x) for x in bounds.reshape(-1)])
    check(load().curobo_hip_mesh_morton_codes(ptr(codes), ptr(v), ptr(f), n, C.cast(b, C.c_void_p), current_stream(v)))
    codes = torch.sort(codes).values.contiguous()
    n_leaves = 1
    while n_leaves * leaf_size < n:
        n_leaves *= 2
    tri = torch.zeros(n, 12, device=device)
    box = torch.zeros(2 * n_leaves, 8, device=device)
    check(load().curobo_hip_mesh_bvh_build(ptr(tri), ptr(box), ptr(v), ptr(f), ptr(codes), n, n_leaves, leaf_size, current_stream(v)))
    s = Mesh(ptr(tri), ptr(box), n, n_leaves, leaf_size, 0)
    return DeviceMesh(tri, box, s, bounds, n)


def mesh_query(mesh: DeviceMesh, points: torch.Tensor, max_distance: float, want_grad: bool = True
               ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """points [n, 3] in the mesh frame -> (sdf [n], gradient [n, 3]) as ``compute_local_sdf_with_grad`` returns them"""
    p = points.to(torch.float32).contiguous()
    n = int(p.shape[0])
    sdf = torch.empty(n, device=p.device)
    grad = torch.empty(n, 3, device=p.device) if want_grad else None
    check(load().curobo_hip_mesh_query(ptr(sdf), ptr(grad), ptr(p), C.addressof(mesh.struct), float(max_distance), n, current_stream(p)))
    return sdf, grad


def mesh_esdf_bake_bvh(out_esdf: torch.Tensor, mesh: DeviceMesh, grid_shape, voxel_size: float, grid_to_mesh_3x4,
                       max_distance: float = 100.0) -> torch.Tensor:
    assert out_esdf.dtype == torch.float16 and out_esdf.is_contiguous()
    nx, ny, nz = (int(v) for v in grid_shape)
    assert out_esdf.numel() == nx * ny * nz
    xf = (C.c_float * 12)(*[float(v) for v in grid_to_mesh_3x4])
    check(load().curobo_hip_mesh_esdf_bake_bvh(ptr(out_esdf), C.addressof(mesh.struct), nx, ny, nz, float(voxel_size), float(max_distance),
                                               C.cast(xf, C.c_void_p), current_stream(out_esdf)))
    return out_esdf


def sphere_mesh_collision(distance, gradient, spheres, mesh_set: MeshSet, weight, activation_distance, env_query_idx, batch_size: int,
                          horizon: int, num_spheres: int, use_multi_env: bool, sweep_steps: int = 0, enable_speed_metric: bool = False,
                          speed_dt=None, accumulate: bool = True):
    """the mesh share of the scene-collision forward (``curobo_hip_sphere_mesh_collision``)"""
    check(load().curobo_hip_sphere_mesh_collision(
        ptr(distance), ptr(gradient), ptr(spheres), C.addressof(mesh_set), ptr(weight), ptr(activation_distance), ptr(env_query_idx),
        batch_size, horizon, num_spheres, int(use_multi_env), sweep_steps, int(enable_speed_metric), ptr(speed_dt), int(accumulate),
        current_stream(distance)))
