"""HIP backend for sphere-vs-scene collision and the per-trajectory cost reduction.

The reference has no backend hook for these: scene collision is NVIDIA Warp code launched from
``curobo/_src/geom/collision/wp_autograd.py:37-249`` and the cost reduction is
``torch.cat`` + ``sum`` (``curobo/_src/rollout/metrics.py:233-265``).  The functions keep the
reference backends' style: pre-allocated tensors in, mutated in place, current stream.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .._lib import Scene, check, current_stream, load, ptr


def make_scene(
    cuboid_dims: Optional[torch.Tensor] = None,
    cuboid_inv_pose: Optional[torch.Tensor] = None,
    cuboid_enable: Optional[torch.Tensor] = None,
    cuboid_count: Optional[torch.Tensor] = None,
    voxel_params: Optional[torch.Tensor] = None,
    voxel_inv_pose: Optional[torch.Tensor] = None,
    voxel_enable: Optional[torch.Tensor] = None,
    voxel_count: Optional[torch.Tensor] = None,
    voxel_features: Optional[torch.Tensor] = None,
    voxel_max_distance: float = 10000.0,
    voxel_coarse_min: Optional[torch.Tensor] = None,
    voxel_coarse_block: int = 0,
    voxel_coarse_dilate: int = 0,
) -> Scene:
    """Pack device tensors into the ``curobo_hip_scene`` struct (host side, plain pointers).

    Layouts are those of the reference stores: ``CuboidData`` (geom/data/data_cuboid.py:67-108:
    dims [E,n,4], inv_pose [E,n,8], enable u8 [E,n], count i32 [E]) and ``VoxelData``
    (geom/data/data_voxel.py:42-95: params [E,n,4], inv_pose [E,n,8], features fp16 [E,n,nvox,1]).
    The caller keeps the tensors alive.
    """
    s = Scene()
    if cuboid_dims is not None and cuboid_dims.numel() > 0:
        assert cuboid_dims.dtype == torch.float32 and cuboid_inv_pose.dtype == torch.float32
        assert cuboid_enable.dtype == torch.uint8 and cuboid_count.dtype == torch.int32
        s.cuboid_dims, s.cuboid_inv_pose = ptr(cuboid_dims), ptr(cuboid_inv_pose)
        s.cuboid_enable, s.cuboid_count = ptr(cuboid_enable), ptr(cuboid_count)
        s.max_cuboids = cuboid_dims.shape[1]
        s.cuboid_has_primitives = int(bool((cuboid_dims[..., 3] != 0).any()))  # host read-back at scene build
    if voxel_params is not None and voxel_params.numel() > 0:
        assert voxel_params.dtype == torch.float32 and voxel_features.dtype == torch.float16
        assert voxel_enable.dtype == torch.uint8 and voxel_count.dtype == torch.int32
        s.voxel_params, s.voxel_inv_pose = ptr(voxel_params), ptr(voxel_inv_pose)
        s.voxel_enable, s.voxel_count = ptr(voxel_enable), ptr(voxel_count)
        s.voxel_features = ptr(voxel_features)
        s.max_voxel_grids = voxel_params.shape[1]
        s.voxel_n_voxels = voxel_features.numel() // (voxel_params.shape[0] * voxel_params.shape[1])
        s.voxel_max_distance = float(voxel_max_distance)
        if voxel_coarse_min is not None:  # optional culling aid, see build_voxel_coarse_min
            assert voxel_coarse_min.dtype == torch.float16 and voxel_coarse_block >= 1 and voxel_coarse_dilate >= 1
            s.voxel_coarse_min = ptr(voxel_coarse_min)
            s.voxel_coarse_block, s.voxel_coarse_dilate = int(voxel_coarse_block), int(voxel_coarse_dilate)
            s.voxel_n_coarse = voxel_coarse_min.numel() // (voxel_params.shape[0] * voxel_params.shape[1])
    return s


def build_voxel_coarse_min(voxel_features: torch.Tensor, voxel_params, block: int = 4, dilate: int = 3) -> torch.Tensor:
    """fp16 [E, n, n_coarse]: per grid the ESDF minimum over every ``block``^3 block of voxels dilated by ``dilate``
    voxels (``curobo_hip_scene.voxel_coarse_min``).  ``voxel_params`` [E, n, 4] = (nx, ny, nz, voxel_size) on the
    host.  A min-pool on the device (scene upload time, not the hot path); fp16 minima of fp16 values are exact."""
    import numpy as np
    import torch.nn.functional as F

    prm = np.asarray(voxel_params, np.float32)
    E, n = prm.shape[0], prm.shape[1]
    feats = voxel_features.reshape(E, n, -1)
    grids, n_coarse = [], 1
    for e in range(E):
        row = []
        for g in range(n):
            nx, ny, nz = (int(v) for v in prm[e, g, :3])
            if nx * ny * nz == 0:  # padding slot (environments with fewer grids): its coarse row stays "never culls"
                row.append(None)
                continue
            x = feats[e, g, : nx * ny * nz].reshape(1, 1, nx, ny, nz).float()
            c = -F.max_pool3d(-x, kernel_size=block + 2 * dilate, stride=block, padding=dilate, ceil_mode=True)
            cx, cy, cz = -(-nx // block), -(-ny // block), -(-nz // block)
            c = c[0, 0, :cx, :cy, :cz].contiguous()
            assert c.shape == (cx, cy, cz)
            row.append(c.reshape(-1).half())
            n_coarse = max(n_coarse, cx * cy * cz)
        grids.append(row)
    out = torch.full((E, n, n_coarse), -65504.0, dtype=torch.float16, device=voxel_features.device)  # padding never culls
    for e in range(E):
        for g in range(n):
            if grids[e][g] is not None:
                out[e, g, : grids[e][g].numel()] = grids[e][g]
    return out


def sphere_obstacle_collision(
    distance: torch.Tensor,
    gradient: torch.Tensor,
    spheres: torch.Tensor,
    scene: Scene,
    weight: torch.Tensor,
    activation_distance: torch.Tensor,
    env_query_idx: Optional[torch.Tensor],
    batch_size: int,
    horizon: int,
    num_spheres: int,
    use_multi_env: bool,
    sweep_steps: int = 0,
    enable_speed_metric: bool = False,
    speed_dt: Optional[torch.Tensor] = None,
):
    """distance[b,h,s], gradient[b,h,s,4] <- activation-shaped penetration of every sphere
    against every enabled obstacle (buffers are fully rewritten, no zero_() needed).  Mesh obstacles (``scene.mesh_set``,
    attached by ``SceneData.from_arrays(..., meshes=)``) are a second launch that adds its share, as the reference launches
    its kernel once per obstacle kind into the same buffers (wp_autograd.py:85-99)."""
    mesh_set = getattr(scene, "mesh_set", None)
    others = scene.max_cuboids > 0 or scene.max_voxel_grids > 0
    if others or mesh_set is None:
        check(load().curobo_hip_sphere_obstacle_collision(
            ptr(distance), ptr(gradient), ptr(spheres), C.addressof(scene), ptr(weight),
            ptr(activation_distance), ptr(env_query_idx), batch_size, horizon, num_spheres,
            int(use_multi_env), sweep_steps, int(enable_speed_metric), ptr(speed_dt),
            current_stream(distance),
        ))
    if mesh_set is not None:
        from .mesh import sphere_mesh_collision

        sphere_mesh_collision(distance, gradient, spheres, mesh_set, weight, activation_distance, env_query_idx, batch_size, horizon,
                              num_spheres, use_multi_env, sweep_steps, enable_speed_metric, speed_dt, accumulate=others)


def trajectory_cost_sum(
    out_cost: torch.Tensor,
    self_cost: Optional[torch.Tensor],
    scene_cost: Optional[torch.Tensor],
    batch_size: int,
    horizon: int,
    num_spheres: int,
):
    """out[b] = sum_h(self[b,h] + sum_s scene[b,h,s]) with one wavefront per trajectory."""
    check(load().curobo_hip_trajectory_cost_sum(
        ptr(out_cost), ptr(self_cost), ptr(scene_cost), batch_size, horizon, num_spheres,
        current_stream(out_cost),
    ))



def mesh_esdf_bake(out_esdf: torch.Tensor, vertices: torch.Tensor, faces: torch.Tensor, grid_shape, voxel_size: float,
                   grid_to_mesh_3x4, max_distance: float = 100.0) -> torch.Tensor:
    """Closed triangle mesh -> fp16 ESDF grid on the device (``curobo_hip_mesh_esdf_bake``; the reference queries
    meshes through Warp's BVH, geom/data/data_mesh.py:555-700).  ``out_esdf`` fp16 [nx * ny * nz] (pre-allocated),
    ``vertices`` f32 [V, 3] in the mesh frame, ``faces`` i32 [F, 3], ``grid_to_mesh_3x4`` 12 floats (host)."""
    assert out_esdf.dtype == torch.float16 and vertices.dtype == torch.float32 and faces.dtype == torch.int32
    assert vertices.is_contiguous() and faces.is_contiguous() and out_esdf.is_contiguous()
    nx, ny, nz = (int(v) for v in grid_shape)
    assert out_esdf.numel() == nx * ny * nz
    xf = (C.c_float * 12)(*[float(v) for v in grid_to_mesh_3x4])
    check(load().curobo_hip_mesh_esdf_bake(ptr(out_esdf), ptr(vertices), ptr(faces), int(vertices.shape[0]), int(faces.shape[0]),
                                           nx, ny, nz, float(voxel_size), float(max_distance), C.cast(xf, C.c_void_p),
                                           current_stream(out_esdf)))
    return out_esdf
