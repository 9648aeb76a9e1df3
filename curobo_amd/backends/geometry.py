"""HIP backend for robot self-collision (reference
``curobo/_src/curobolib/backends/cuda_core_backend/geometry.py:63-227``,
``pybind/geometry_bindings.cpp:16-45``)."""

from __future__ import annotations

import torch

from .._lib import check, current_stream, load, ptr


def self_collision_distance(
    out_distance: torch.Tensor,
    out_vec: torch.Tensor,
    pair_distance: torch.Tensor,
    sparse_index: torch.Tensor,
    robot_spheres: torch.Tensor,
    sphere_padding: torch.Tensor,
    weight: torch.Tensor,
    pair_locations: torch.Tensor,
    block_batch_max_value: torch.Tensor,
    block_batch_max_index: torch.Tensor,
    num_blocks_per_batch: int,
    max_threads_per_block: int,
    batch_size: int,
    horizon: int,
    nspheres: int,
    num_collision_pairs: int,
    store_pair_distance: bool,
    compute_grad: bool,
):
    """Max sphere-pair penetration per point; modifies the output tensors in place."""
    check(load().curobo_hip_self_collision_distance(
        ptr(out_distance), ptr(out_vec), ptr(pair_distance), ptr(sparse_index), ptr(robot_spheres),
        ptr(sphere_padding), ptr(weight), ptr(pair_locations), ptr(block_batch_max_value),
        ptr(block_batch_max_index), num_blocks_per_batch, max_threads_per_block, batch_size,
        horizon, nspheres, num_collision_pairs, int(store_pair_distance), int(compute_grad),
        current_stream(out_distance),
    ))
