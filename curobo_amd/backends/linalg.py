"""HIP backend for the dense linear algebra of the seed-IK solver.

The reference runs the Levenberg-Marquardt step as an NVIDIA Warp tile kernel without a backend
hook (``curobo/_src/optim/util/levenberg_marquardt_step.py:96-199``); this is the replacement body of
``LevenbergMarquardtStep.__call__`` (same tensors, mutated in place, current stream)."""

from __future__ import annotations

import torch

from .._lib import check, current_stream, load, ptr


def levenberg_marquardt_step(
    joint_position_out: torch.Tensor,
    pred_reduction: torch.Tensor,
    jacobian: torch.Tensor,
    jTerror: torch.Tensor,
    lambda_damping: torch.Tensor,
    joint_position_in: torch.Tensor,
):
    """jacobian [b, n_residuals, action_dim]; J^T J on the matrix cores, Cholesky solve in LDS."""
    b, r, d = jacobian.shape
    check(load().curobo_hip_levenberg_marquardt_step(
        ptr(joint_position_out), ptr(pred_reduction), ptr(jacobian), ptr(jTerror), ptr(lambda_damping),
        ptr(joint_position_in), b, r, d, current_stream(joint_position_out),
    ))


def seed_ik_update_state(
    joint_position, jacobian, jTerror, error_norm, position_error, orientation_error, lambda_damping, success,
    improvement, candidate_joint_position, candidate_pose_jacobian, candidate_pose_jTerror, candidate_pose_cost,
    candidate_position_distance, candidate_rotation_distance, predicted_reduction, action_min, action_max,
    current_position, dt, velocity_limits, joint_limit_weight: float, rho_min: float, lambda_factor: float,
    lambda_min: float, lambda_max: float, convergence_position_tolerance: float,
    convergence_orientation_tolerance: float, convergence_joint_limit_weight: float, initial: bool,
    current_velocity=None, velocity_weight: float = 0.0, acceleration_weight: float = 0.0,
):
    """One launch for the torch glue of a seed-IK LM iteration (``curobo_hip_seed_ik_update_state``):
    joint-limit residual rows + trust ratio + acceptance + damping + state selection + convergence."""
    n, d = candidate_joint_position.shape
    t = candidate_position_distance.shape[-1]
    check(load().curobo_hip_seed_ik_update_state(
        ptr(joint_position), ptr(jacobian), ptr(jTerror), ptr(error_norm), ptr(position_error), ptr(orientation_error),
        ptr(lambda_damping), ptr(success), ptr(improvement), ptr(candidate_joint_position), ptr(candidate_pose_jacobian),
        ptr(candidate_pose_jTerror), ptr(candidate_pose_cost), ptr(candidate_position_distance),
        ptr(candidate_rotation_distance), ptr(predicted_reduction), ptr(action_min), ptr(action_max), ptr(current_position),
        ptr(dt), ptr(velocity_limits), ptr(current_velocity), float(velocity_weight), float(acceleration_weight),
        float(joint_limit_weight), float(rho_min), float(lambda_factor), float(lambda_min),
        float(lambda_max), float(convergence_position_tolerance), float(convergence_orientation_tolerance),
        float(convergence_joint_limit_weight), n, d, t, int(initial), current_stream(joint_position),
    ))


def seed_ik_iterate(
    joint_position, jacobian, jTerror, error_norm, position_error, orientation_error, lambda_damping, success, improvement,
    seed_joint_position, goal_position, goal_quat, idxs_goal, position_orientation_weight, pose_axes_weight_factor,
    pose_convergence_tolerance, project_distance_to_goal, num_goalset: int, rotation_method: int, fixed_transform,
    joint_map_type, joint_map, link_map, tool_frame_map, link_chain_data, link_chain_offsets, joint_links_data,
    joint_links_offsets, joint_affects_endeffector, joint_offset_map, action_min, action_max, current_position, dt,
    velocity_limits, joint_limit_weight: float, rho_min: float, lambda_factor: float, lambda_min: float, lambda_max: float,
    convergence_position_tolerance: float, convergence_orientation_tolerance: float, convergence_joint_limit_weight: float,
    iterations: int, initial: bool, current_velocity=None, velocity_weight: float = 0.0, acceleration_weight: float = 0.0,
    stop_flag=None, blocks_run=None,
):
    """``iterations`` whole LM iterations of the seed-IK solver (plus the initial evaluation of ``seed_joint_position``
    when ``initial``) in one launch (``curobo_hip_seed_ik_iterate``); state buffers as in :func:`seed_ik_update_state`."""
    n, d = joint_position.shape
    t = int(tool_frame_map.shape[0])
    check(load().curobo_hip_seed_ik_iterate(
        ptr(joint_position), ptr(jacobian), ptr(jTerror), ptr(error_norm), ptr(position_error), ptr(orientation_error),
        ptr(lambda_damping), ptr(success), ptr(improvement), ptr(seed_joint_position), ptr(goal_position), ptr(goal_quat),
        ptr(idxs_goal), ptr(position_orientation_weight), ptr(pose_axes_weight_factor), ptr(pose_convergence_tolerance),
        ptr(project_distance_to_goal), int(num_goalset), int(rotation_method), ptr(fixed_transform), ptr(joint_map_type),
        ptr(joint_map), ptr(link_map), ptr(tool_frame_map), ptr(link_chain_data), ptr(link_chain_offsets),
        ptr(joint_links_data), ptr(joint_links_offsets), ptr(joint_affects_endeffector), ptr(joint_offset_map),
        ptr(action_min), ptr(action_max), ptr(current_position), ptr(dt), ptr(velocity_limits), ptr(current_velocity),
        float(velocity_weight), float(acceleration_weight), float(joint_limit_weight), float(rho_min), float(lambda_factor),
        float(lambda_min), float(lambda_max), float(convergence_position_tolerance), float(convergence_orientation_tolerance),
        float(convergence_joint_limit_weight), n, d, int(link_map.shape[0]), t, int(link_chain_data.shape[0]), int(iterations),
        int(initial), ptr(stop_flag), ptr(blocks_run), current_stream(joint_position),
    ))


def seed_ik_iterate_fits(dof: int, num_links: int, num_tool_frames: int, link_chain_len: int) -> bool:
    """whether :func:`seed_ik_iterate` can hold 16 problems' state in its 64 KB of LDS (``curobo_hip_seed_ik_iterate_fits``)"""
    return bool(load().curobo_hip_seed_ik_iterate_fits(int(dof), int(num_links), int(num_tool_frames), int(link_chain_len)))


def seed_ik_batch_status(success, num_problems: int, num_seeds: int, needed: int, stop_flag):
    """device-side exit test of the seed-IK solver (``curobo_hip_seed_ik_batch_status``)"""
    check(load().curobo_hip_seed_ik_batch_status(ptr(success), int(num_problems), int(num_seeds), int(needed), ptr(stop_flag),
                                                 current_stream(success)))


def seed_ik_select(out_success, out_solution, out_position_error, out_orientation_error, joint_position, position_error,
                   orientation_error, limit_lower, limit_upper, current_position, position_tolerance: float,
                   orientation_tolerance: float, start_cspace_dist_weight: float, check_limits: bool, return_seeds: int):
    """the ``return_seeds`` best seeds of every problem, best first (``curobo_hip_seed_ik_select``)"""
    p, s, d = joint_position.shape
    check(load().curobo_hip_seed_ik_select(
        ptr(out_success), ptr(out_solution), ptr(out_position_error), ptr(out_orientation_error), ptr(joint_position),
        ptr(position_error), ptr(orientation_error), ptr(limit_lower), ptr(limit_upper), ptr(current_position),
        float(position_tolerance), float(orientation_tolerance), float(start_cspace_dist_weight), int(check_limits), p, s, d,
        int(return_seeds), current_stream(joint_position)))


def ik_rank(out_success, out_solution, out_position_error, out_rotation_error, out_cost, out_seed_index, out_goalset_index,
            joint_position, cost, position_distance, rotation_distance, self_collision_distance, cspace_cost, scene_distance,
            goalset_idx, position_threshold: float, rotation_threshold: float, num_problems: int, num_seeds: int,
            return_seeds: int, seed_offset: int):
    """feasibility, success and the ranked winners of an IK batch in one launch (``curobo_hip_ik_rank``)"""
    d = joint_position.shape[-1]
    t = position_distance.shape[-1]
    n_scene = 0 if scene_distance is None else scene_distance.numel() // (num_problems * num_seeds)
    check(load().curobo_hip_ik_rank(
        ptr(out_success), ptr(out_solution), ptr(out_position_error), ptr(out_rotation_error), ptr(out_cost),
        ptr(out_seed_index), ptr(out_goalset_index), ptr(joint_position), ptr(cost), ptr(position_distance),
        ptr(rotation_distance), ptr(self_collision_distance), ptr(cspace_cost), ptr(scene_distance), ptr(goalset_idx),
        float(position_threshold), float(rotation_threshold), int(num_problems), int(num_seeds), d, t, int(n_scene),
        int(return_seeds), int(seed_offset), current_stream(joint_position)))


def argmin_rows(out_rows, cost, payload, seed_offset: int):
    """(min cost, global index of the first seed that attains it, its payload row) per problem (``curobo_hip_argmin_rows``)"""
    p, s = cost.shape
    check(load().curobo_hip_argmin_rows(ptr(out_rows), ptr(cost), ptr(payload), p, s, int(payload.shape[-1]), int(seed_offset),
                                        current_stream(cost)))
