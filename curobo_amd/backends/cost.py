"""HIP backend for the tool-pose and c-space POSITION cost kernels.

The reference runs these through NVIDIA Warp without a backend hook
(``curobo/_src/cost/wp_tool_pose.py:698-914``, ``cost/wp_cspace_position.py:232-362``); the
functions keep the reference backends' style: pre-allocated tensors in, mutated in place, launched
on the current stream.  Argument order = the Warp kernels' input order.
"""

from __future__ import annotations

from typing import Optional

import torch

from .._lib import check, current_stream, load, ptr


def tool_pose_distance(
    out_distance: torch.Tensor,
    out_position_distance: torch.Tensor,
    out_rotation_distance: torch.Tensor,
    out_position_gradient: torch.Tensor,
    out_rotation_gradient: torch.Tensor,
    out_goalset_idx: torch.Tensor,
    current_position: torch.Tensor,
    current_quat: torch.Tensor,
    goal_position: torch.Tensor,
    goal_quat: torch.Tensor,
    idxs_goal: torch.Tensor,
    position_orientation_weight: torch.Tensor,
    terminal_pose_axes_weight_factor: torch.Tensor,
    non_terminal_pose_axes_weight_factor: torch.Tensor,
    terminal_pose_convergence_tolerance: torch.Tensor,
    non_terminal_pose_convergence_tolerance: torch.Tensor,
    project_distance_to_goal: torch.Tensor,
    batch_size: int,
    horizon: int,
    num_links: int,
    num_goalset: int,
    rotation_method: int = 0,
):
    check(load().curobo_hip_tool_pose_distance(
        ptr(out_distance), ptr(out_position_distance), ptr(out_rotation_distance), ptr(out_position_gradient),
        ptr(out_rotation_gradient), ptr(out_goalset_idx), ptr(current_position), ptr(current_quat),
        ptr(goal_position), ptr(goal_quat), ptr(idxs_goal), ptr(position_orientation_weight),
        ptr(terminal_pose_axes_weight_factor), ptr(non_terminal_pose_axes_weight_factor),
        ptr(terminal_pose_convergence_tolerance), ptr(non_terminal_pose_convergence_tolerance),
        ptr(project_distance_to_goal), batch_size, horizon, num_links, num_goalset, rotation_method,
        current_stream(out_distance),
    ))


def cspace_position_cost(
    out_cost: torch.Tensor,
    out_grad_p: torch.Tensor,
    out_grad_tau: Optional[torch.Tensor],
    pos: torch.Tensor,
    effort: Optional[torch.Tensor],
    cspace_target: torch.Tensor,
    cspace_target_idx: torch.Tensor,
    p_b: torch.Tensor,
    effort_b: torch.Tensor,
    weight: torch.Tensor,
    activation_distance: torch.Tensor,
    cspace_target_weight: torch.Tensor,
    cspace_target_dof_weight: torch.Tensor,
    squared_l2_reg_weight: torch.Tensor,
    current_position: torch.Tensor,
    current_velocity: torch.Tensor,
    idxs_current_state: torch.Tensor,
    v_b: torch.Tensor,
    state_dt: torch.Tensor,
    write_grad: bool,
    batch_size: int,
    horizon: int,
    dof: int,
):
    check(load().curobo_hip_cspace_position_cost(
        ptr(out_cost), ptr(out_grad_p), ptr(out_grad_tau), ptr(pos), ptr(effort), ptr(cspace_target),
        ptr(cspace_target_idx), ptr(p_b), ptr(effort_b), ptr(weight), ptr(activation_distance),
        ptr(cspace_target_weight), ptr(cspace_target_dof_weight), ptr(squared_l2_reg_weight),
        ptr(current_position), ptr(current_velocity), ptr(idxs_current_state), ptr(v_b), ptr(state_dt),
        int(write_grad), batch_size, horizon, dof, current_stream(out_cost),
    ))


def rollout_point_aggregate(
    out_cost: torch.Tensor,
    grad_q: Optional[torch.Tensor],
    pose_cost: Optional[torch.Tensor],
    cspace_cost: Optional[torch.Tensor],
    cspace_grad: Optional[torch.Tensor],
    self_cost: Optional[torch.Tensor],
    scene_cost: Optional[torch.Tensor],
    rows: int,
    num_links: int,
    dof: int,
    num_spheres: int,
):
    """out_cost[r] = all cost terms of rollout row r summed; grad_q[r] += cspace_grad[r]."""
    check(load().curobo_hip_rollout_point_aggregate(
        ptr(out_cost), ptr(grad_q), ptr(pose_cost), ptr(cspace_cost), ptr(cspace_grad), ptr(self_cost),
        ptr(scene_cost), rows, num_links, dof, num_spheres, current_stream(out_cost),
    ))


def cspace_state_cost(
    out_cost, out_grad_p, out_grad_v, out_grad_a, out_grad_j, out_grad_tau, pos, vel, acc, jerk, effort, state_dt,
    target_joint_position, idxs_target_joint_position, p_b, v_b, a_b, j_b, effort_b, weight, activation_distance,
    squared_l2_regularization_weights, cspace_target_weight, cspace_non_terminal_weight_factor,
    cspace_target_dof_weight, write_grad: bool, batch_size: int, horizon: int, dof: int,
    retime_weights: bool = False, retime_regularization_weights: bool = False,
):
    """reference ``forward_cspace_state_warp`` (``cost/wp_cspace_state.py:20-287``), argument order =
    the Warp kernel's inputs then outputs-first like the other backends."""
    check(load().curobo_hip_cspace_state_cost(
        ptr(out_cost), ptr(out_grad_p), ptr(out_grad_v), ptr(out_grad_a), ptr(out_grad_j), ptr(out_grad_tau), ptr(pos),
        ptr(vel), ptr(acc), ptr(jerk), ptr(effort), ptr(state_dt), ptr(target_joint_position),
        ptr(idxs_target_joint_position), ptr(p_b), ptr(v_b), ptr(a_b), ptr(j_b), ptr(effort_b), ptr(weight),
        ptr(activation_distance), ptr(squared_l2_regularization_weights), ptr(cspace_target_weight),
        ptr(cspace_non_terminal_weight_factor), ptr(cspace_target_dof_weight), int(write_grad), batch_size, horizon, dof,
        int(retime_weights), int(retime_regularization_weights), current_stream(out_cost),
    ))


def cspace_l2_distance(out_cost, out_grad_p, pos, target, target_idx, weight, terminal_dof_weight,
                       non_terminal_dof_weight, write_grad: bool, batch_size: int, horizon: int, dof: int):
    """reference ``forward_l2_warp`` (``cost/wp_torch_cspace_dist.py:12-78``; outputs first)."""
    check(load().curobo_hip_cspace_l2_distance(
        ptr(out_cost), ptr(out_grad_p), ptr(pos), ptr(target), ptr(target_idx), ptr(weight), ptr(terminal_dof_weight),
        ptr(non_terminal_dof_weight), int(write_grad), batch_size, horizon, dof, current_stream(out_cost),
    ))
