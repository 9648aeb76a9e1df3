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
    if voxel_params is not None and voxel_params.numel() > 0:
        assert voxel_params.dtype == torch.float32 and voxel_features.dtype == torch.float16
        assert voxel_enable.dtype == torch.uint8 and voxel_count.dtype == torch.int32
        s.voxel_params, s.voxel_inv_pose = ptr(voxel_params), ptr(voxel_inv_pose)
        s.voxel_enable, s.voxel_count = ptr(voxel_enable), ptr(voxel_count)
        s.voxel_features = ptr(voxel_features)
        s.max_voxel_grids = voxel_params.shape[1]
        s.voxel_n_voxels = voxel_features.numel() // (voxel_params.shape[0] * voxel_params.shape[1])
        s.voxel_max_distance = float(voxel_max_distance)
    return s


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
    against every enabled obstacle (buffers are fully rewritten, no zero_() needed)."""
    check(load().curobo_hip_sphere_obstacle_collision(
        ptr(distance), ptr(gradient), ptr(spheres), C.addressof(scene), ptr(weight),
        ptr(activation_distance), ptr(env_query_idx), batch_size, horizon, num_spheres,
        int(use_multi_env), sweep_steps, int(enable_speed_metric), ptr(speed_dt),
        current_stream(distance),
    ))


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
