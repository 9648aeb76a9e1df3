"""HIP backend for the kinematics kernels.

Function names and positional argument order are those of the reference backends
(``curobo/_src/curobolib/backends/cuda_core_backend/kinematics.py:21-379`` and
``pybind/kinematics_bindings.cpp:128-237``); all tensors are pre-allocated by the caller and
mutated in place; launches go to ``torch.cuda.current_stream()``.
"""

from __future__ import annotations

from typing import Optional

import torch

from .._lib import check, current_stream, load, ptr


def launch_kinematics_forward(
    link_pos: torch.Tensor,
    link_quat: torch.Tensor,
    batch_center_of_mass: torch.Tensor,
    global_cumul_mat: torch.Tensor,
    joint_vec: torch.Tensor,
    fixed_transform: torch.Tensor,
    link_masses_com: torch.Tensor,
    joint_map_type: torch.Tensor,
    joint_map: torch.Tensor,
    link_map: torch.Tensor,
    tool_frame_map: torch.Tensor,
    joint_offset_map: torch.Tensor,
    batch_size: int,
    horizon: int,
    n_joints: int,
    compute_com: bool = False,
):
    """Forward kinematics without sphere output (reference :21-88)."""
    check(load().curobo_hip_launch_kinematics_forward(
        ptr(link_pos), ptr(link_quat), ptr(batch_center_of_mass), ptr(global_cumul_mat),
        ptr(joint_vec), ptr(fixed_transform), ptr(link_masses_com), ptr(joint_map_type),
        ptr(joint_map), ptr(link_map), ptr(tool_frame_map), ptr(joint_offset_map),
        batch_size, horizon, n_joints, link_map.shape[0], tool_frame_map.shape[0],
        int(compute_com), current_stream(joint_vec),
    ))


def launch_kinematics_forward_spheres(
    link_pos: torch.Tensor,
    link_quat: torch.Tensor,
    batch_robot_spheres: torch.Tensor,
    batch_center_of_mass: torch.Tensor,
    global_cumul_mat: torch.Tensor,
    joint_vec: torch.Tensor,
    fixed_transform: torch.Tensor,
    robot_spheres: torch.Tensor,
    link_masses_com: torch.Tensor,
    joint_map_type: torch.Tensor,
    joint_map: torch.Tensor,
    link_map: torch.Tensor,
    tool_frame_map: torch.Tensor,
    link_sphere_map: torch.Tensor,
    joint_offset_map: torch.Tensor,
    env_query_idx: torch.Tensor,
    num_envs: int,
    batch_size: int,
    horizon: int,
    n_joints: int,
    num_spheres: int,
    output_threads_per_batch: int,
    write_global_cumul: bool = True,
    compute_com: bool = False,
):
    """Forward kinematics with sphere output (reference :91-190)."""
    if output_threads_per_batch not in (32, 64, 128):
        raise ValueError("output_threads_per_batch must be one of 32, 64, or 128")
    check(load().curobo_hip_launch_kinematics_forward_spheres(
        ptr(link_pos), ptr(link_quat), ptr(batch_robot_spheres), ptr(batch_center_of_mass),
        ptr(global_cumul_mat), ptr(joint_vec), ptr(fixed_transform), ptr(robot_spheres),
        ptr(link_masses_com), ptr(joint_map_type), ptr(joint_map), ptr(link_map),
        ptr(tool_frame_map), ptr(link_sphere_map), ptr(joint_offset_map), ptr(env_query_idx),
        num_envs, batch_size, horizon, n_joints, num_spheres, link_map.shape[0],
        tool_frame_map.shape[0], int(write_global_cumul), int(compute_com),
        current_stream(joint_vec),
    ))


def launch_kinematics_forward_spheres_jacobian(
    link_pos: torch.Tensor,
    link_quat: torch.Tensor,
    batch_robot_spheres: torch.Tensor,
    batch_center_of_mass: torch.Tensor,
    batch_jacobian: torch.Tensor,
    global_cumul_mat: torch.Tensor,
    joint_vec: torch.Tensor,
    fixed_transform: torch.Tensor,
    robot_spheres: torch.Tensor,
    link_masses_com: torch.Tensor,
    joint_map_type: torch.Tensor,
    joint_map: torch.Tensor,
    link_map: torch.Tensor,
    tool_frame_map: torch.Tensor,
    link_sphere_map: torch.Tensor,
    link_chain_data: torch.Tensor,
    link_chain_offsets: torch.Tensor,
    joint_links_data: torch.Tensor,
    joint_links_offsets: torch.Tensor,
    joint_affects_endeffector: torch.Tensor,
    joint_offset_map: torch.Tensor,
    env_query_idx: torch.Tensor,
    num_envs: int,
    batch_size: int,
    horizon: int,
    n_joints: int,
    num_spheres: int,
    output_threads_per_batch: int,
    write_global_cumul: bool = True,
    compute_com: bool = False,
):
    """Forward kinematics with spheres and geometric Jacobian (reference :193-300)."""
    if output_threads_per_batch not in (32, 64, 128):
        raise ValueError("output_threads_per_batch must be one of 32, 64, or 128")
    check(load().curobo_hip_launch_kinematics_forward_spheres_jacobian(
        ptr(link_pos), ptr(link_quat), ptr(batch_robot_spheres), ptr(batch_center_of_mass),
        ptr(batch_jacobian), ptr(global_cumul_mat), ptr(joint_vec), ptr(fixed_transform),
        ptr(robot_spheres), ptr(link_masses_com), ptr(joint_map_type), ptr(joint_map),
        ptr(link_map), ptr(tool_frame_map), ptr(link_sphere_map), ptr(link_chain_data),
        ptr(link_chain_offsets), ptr(joint_links_data), ptr(joint_links_offsets),
        ptr(joint_affects_endeffector), ptr(joint_offset_map), ptr(env_query_idx),
        num_envs, batch_size, horizon, n_joints, num_spheres, link_map.shape[0],
        tool_frame_map.shape[0], int(write_global_cumul), int(compute_com),
        current_stream(joint_vec),
    ))


def launch_kinematics_backward(
    grad_out: torch.Tensor,
    grad_nlinks_pos: torch.Tensor,
    grad_nlinks_quat: torch.Tensor,
    grad_spheres: torch.Tensor,
    grad_center_of_mass: torch.Tensor,
    batch_center_of_mass: torch.Tensor,
    grad_jacobian: torch.Tensor,
    global_cumul_mat: torch.Tensor,
    robot_spheres: torch.Tensor,
    link_masses_com: torch.Tensor,
    link_map: torch.Tensor,
    joint_map: torch.Tensor,
    joint_map_type: torch.Tensor,
    tool_frame_map: torch.Tensor,
    link_sphere_map: torch.Tensor,
    link_chain_data: torch.Tensor,
    link_chain_offsets: torch.Tensor,
    joint_links_data: torch.Tensor,
    joint_links_offsets: torch.Tensor,
    joint_affects_endeffector: torch.Tensor,
    joint_offset_map: torch.Tensor,
    env_query_idx: torch.Tensor,
    num_envs: int,
    batch_size: int,
    horizon: int,
    n_joints: int,
    num_spheres: int,
    compute_com: bool = False,
    compute_jacobian_grad: bool = False,
    grad_spheres_b: Optional[torch.Tensor] = None,
):
    """Kinematics VJP (reference :303-379).

    ``grad_spheres_b`` (keyword-only extension) is a second sphere-gradient buffer summed on the
    fly, e.g. the scene-collision gradient next to the self-collision gradient.
    """
    check(load().curobo_hip_launch_kinematics_backward(
        ptr(grad_out), ptr(grad_nlinks_pos), ptr(grad_nlinks_quat),
        ptr(grad_spheres) if num_spheres > 0 else None, ptr(grad_spheres_b),
        ptr(grad_center_of_mass), ptr(batch_center_of_mass), ptr(grad_jacobian),
        ptr(global_cumul_mat), ptr(robot_spheres), ptr(link_masses_com), ptr(link_map),
        ptr(joint_map), ptr(joint_map_type), ptr(tool_frame_map), ptr(link_sphere_map),
        ptr(link_chain_data), ptr(link_chain_offsets), ptr(joint_links_data),
        ptr(joint_links_offsets), ptr(joint_affects_endeffector), ptr(joint_offset_map),
        ptr(env_query_idx), num_envs, batch_size, horizon, n_joints, num_spheres,
        link_map.shape[0], tool_frame_map.shape[0], link_chain_data.shape[0], int(compute_com),
        int(compute_jacobian_grad), current_stream(grad_out),
    ))
