"""HIP backend for the fused rollout kernel (``csrc/rollout_fused.hip``).

The reference has no single entry point for this: it is the launch sequence of
``RobotRollout.evaluate_action`` + ``cost.backward`` (``curobo/_src/rollout/rollout_robot.py:252-263,
537-587``; ``optim/components/gradient_opt_core.py:445-480``) -- B-spline forward, FK forward, self
collision, sphere-obstacle collision, cost sum, FK backward, B-spline backward.  Same conventions
as the other backends: pre-allocated tensors in, mutated in place, current stream, no allocation.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .._lib import Scene, TrajOptTerms, check, current_stream, load, ptr


def _rollout_trajectory(fn, terms, with_terms,

    out_cost: torch.Tensor,
    out_grad_knots: torch.Tensor,
    out_position: Optional[torch.Tensor],
    out_robot_spheres: Optional[torch.Tensor],
    u_position: torch.Tensor,
    start_position: torch.Tensor,
    start_velocity: torch.Tensor,
    start_acceleration: torch.Tensor,
    start_jerk: torch.Tensor,
    goal_position: torch.Tensor,
    goal_velocity: torch.Tensor,
    goal_acceleration: torch.Tensor,
    goal_jerk: torch.Tensor,
    start_idx: torch.Tensor,
    goal_idx: torch.Tensor,
    traj_dt: torch.Tensor,
    use_implicit_goal_state: torch.Tensor,
    fixed_transform: torch.Tensor,
    robot_spheres: torch.Tensor,
    joint_map_type: torch.Tensor,
    joint_map: torch.Tensor,
    link_map: torch.Tensor,
    link_sphere_map: torch.Tensor,
    link_chain_data: torch.Tensor,
    link_chain_offsets: torch.Tensor,
    joint_offset_map: torch.Tensor,
    sphere_padding: Optional[torch.Tensor],
    self_collision_weight: Optional[torch.Tensor],
    pair_locations: Optional[torch.Tensor],
    scene: Optional[Scene],
    scene_collision_weight: Optional[torch.Tensor],
    activation_distance: Optional[torch.Tensor],
    speed_dt: Optional[torch.Tensor],
    env_query_idx: torch.Tensor,
    num_envs: int,
    use_multi_env: bool,
    batch_size: int,
    padded_horizon: int,
    dof: int,
    n_knots: int,
    bspline_degree: int,
    sweep_steps: int = 0,
    enable_speed_metric: bool = False,
    dispatch: Optional["DispatchOrder"] = None,
):
    num_pairs = 0 if pair_locations is None else int(pair_locations.shape[0])
    lanes = getattr(pair_locations, "_self_lane_lists", None)
    ws, phase = (None, 0) if dispatch is None else dispatch.next(batch_size)
    extra = ((None if terms is None else C.addressof(terms)),) if with_terms else ()
    check(fn(
        ptr(out_cost), ptr(out_grad_knots), ptr(out_position), ptr(out_robot_spheres), ptr(u_position),
        ptr(start_position), ptr(start_velocity), ptr(start_acceleration), ptr(start_jerk),
        ptr(goal_position), ptr(goal_velocity), ptr(goal_acceleration), ptr(goal_jerk), ptr(start_idx),
        ptr(goal_idx), ptr(traj_dt), ptr(use_implicit_goal_state), ptr(fixed_transform), ptr(robot_spheres),
        ptr(joint_map_type), ptr(joint_map), ptr(link_map), ptr(link_sphere_map), ptr(link_chain_data),
        ptr(link_chain_offsets), ptr(joint_offset_map), ptr(sphere_padding), ptr(self_collision_weight),
        ptr(pair_locations), None if scene is None else C.addressof(scene), ptr(scene_collision_weight),
        ptr(activation_distance), ptr(speed_dt), ptr(env_query_idx), num_envs, int(use_multi_env),
        batch_size, padded_horizon, dof, n_knots, bspline_degree, int(fixed_transform.shape[0]),
        int(link_sphere_map.shape[0]), num_pairs, int(link_chain_data.shape[0]), sweep_steps,
        int(enable_speed_metric), ptr(ws), phase, ptr(lanes[0]) if lanes else None, lanes[1] if lanes else 0, *extra,
        current_stream(out_cost),
    ))


def attach_self_lane_lists(pair_locations: Optional[torch.Tensor], num_spheres: int) -> None:
    """Deal the robot's collision pairs to lanes once (``curobo_hip_self_lane_lists_host``) and hang the device copy on the
    pair tensor itself: the fused trajectory launches find it there (``_self_lane_lists``) and run the lane = sphere form of
    the self-collision pass; a pair tensor without it (a copy, a slice) runs the row form.  One device -> host read of the
    pair list: call at set-up time, never inside a captured launch sequence."""
    if pair_locations is None or pair_locations.numel() == 0 or getattr(pair_locations, "_self_lane_lists", None) is not None:
        return
    host = pair_locations.detach().to("cpu", torch.int16).contiguous()
    n_pairs = int(host.shape[0])
    cap = 64 * (n_pairs // 64 + 2) * (2 if num_spheres > 64 else 1) + 64 * 64
    out = torch.zeros(cap, dtype=torch.int32)
    code = int(load().curobo_hip_self_lane_lists_host(out.data_ptr(), cap, host.data_ptr(), n_pairs, int(num_spheres)))
    if code < 0:
        check(code)
    if code == 0:
        return
    n = ((code & 0xffff) + (code >> 16)) * 64
    pair_locations._self_lane_lists = (out[:n].to(pair_locations.device).contiguous(), code)



def rollout_trajectory_fused(
    out_cost: torch.Tensor,
    out_grad_knots: torch.Tensor,
    out_position: Optional[torch.Tensor],
    out_robot_spheres: Optional[torch.Tensor],
    u_position: torch.Tensor,
    start_position: torch.Tensor,
    start_velocity: torch.Tensor,
    start_acceleration: torch.Tensor,
    start_jerk: torch.Tensor,
    goal_position: torch.Tensor,
    goal_velocity: torch.Tensor,
    goal_acceleration: torch.Tensor,
    goal_jerk: torch.Tensor,
    start_idx: torch.Tensor,
    goal_idx: torch.Tensor,
    traj_dt: torch.Tensor,
    use_implicit_goal_state: torch.Tensor,
    fixed_transform: torch.Tensor,
    robot_spheres: torch.Tensor,
    joint_map_type: torch.Tensor,
    joint_map: torch.Tensor,
    link_map: torch.Tensor,
    link_sphere_map: torch.Tensor,
    link_chain_data: torch.Tensor,
    link_chain_offsets: torch.Tensor,
    joint_offset_map: torch.Tensor,
    sphere_padding: Optional[torch.Tensor],
    self_collision_weight: Optional[torch.Tensor],
    pair_locations: Optional[torch.Tensor],
    scene: Optional[Scene],
    scene_collision_weight: Optional[torch.Tensor],
    activation_distance: Optional[torch.Tensor],
    speed_dt: Optional[torch.Tensor],
    env_query_idx: torch.Tensor,
    num_envs: int,
    use_multi_env: bool,
    batch_size: int,
    padded_horizon: int,
    dof: int,
    n_knots: int,
    bspline_degree: int,
    sweep_steps: int = 0,
    enable_speed_metric: bool = False,
    dispatch: Optional["DispatchOrder"] = None,
):
    """cost[b], grad_knots[b, n_knots, dof] of ``batch_size`` B-spline trajectories in one launch."""
    return _rollout_trajectory(load().curobo_hip_rollout_trajectory_fused, None, False, out_cost, out_grad_knots, out_position, out_robot_spheres, u_position, start_position, start_velocity, start_acceleration, start_jerk, goal_position, goal_velocity, goal_acceleration, goal_jerk, start_idx, goal_idx, traj_dt, use_implicit_goal_state, fixed_transform, robot_spheres, joint_map_type, joint_map, link_map, link_sphere_map, link_chain_data, link_chain_offsets, joint_offset_map, sphere_padding, self_collision_weight, pair_locations, scene, scene_collision_weight, activation_distance, speed_dt, env_query_idx, num_envs, use_multi_env, batch_size, padded_horizon, dof, n_knots, bspline_degree, sweep_steps, enable_speed_metric, dispatch)


FUSED_LDS_LIMIT = 160 * 1024


class DispatchOrder:
    """Longest-first dispatch workspace of the fused trajectory kernels (``dispatch_ws`` /
    ``dispatch_phase`` of the header): one per rollout instance, i.e. per sequence of launches on the
    same batch of optimisation variables.  ``next`` hands out the workspace with alternating phase
    (also under graph capture: the captured launches keep the phase they were captured with; any
    phase sequence is safe, alternation just keeps the duration estimates one launch old)."""

    def __init__(self, batch_size: int, device: torch.device):
        lib = load()
        self.batch_size = batch_size
        self.ws = torch.empty(int(lib.curobo_hip_rollout_dispatch_ws_size(batch_size)), dtype=torch.int32, device=device)
        check(lib.curobo_hip_rollout_dispatch_ws_init(ptr(self.ws), batch_size, current_stream(self.ws)))
        self._phase = 1

    def next(self, batch_size: int):
        if batch_size != self.batch_size:
            raise ValueError(f"dispatch workspace was made for batch {self.batch_size}, launch has {batch_size}")
        self._phase ^= 1
        return self.ws, self._phase


def rollout_trajectory_fused_lds_bytes(padded_horizon: int, dof: int, num_links: int, num_spheres: int,
                                       num_collision_pairs: int, link_chain_len: int, num_obstacles: int) -> int:
    """LDS bytes one trajectory needs in the fused kernel (usable when <= FUSED_LDS_LIMIT)."""
    return int(load().curobo_hip_rollout_trajectory_fused_lds_bytes(
        padded_horizon, dof, num_links, num_spheres, num_collision_pairs, link_chain_len, num_obstacles))


def rollout_ik_fused(
    out_cost, out_grad_q, out_pose_distance, out_position_distance, out_rotation_distance, out_goalset_idx,
    out_link_pos, out_link_quat, out_robot_spheres, out_cspace_cost, q, goal_position, goal_quat, idxs_goal,
    position_orientation_weight, terminal_pose_axes_weight_factor, terminal_pose_convergence_tolerance,
    project_distance_to_goal, num_goalset: int, rotation_method: int, p_b, cspace_weight, cspace_activation_distance,
    fixed_transform, robot_spheres, joint_map_type, joint_map, link_map, tool_frame_map, link_sphere_map,
    link_chain_data, link_chain_offsets, joint_offset_map, sphere_padding, self_collision_weight, pair_locations,
    scene: Optional[Scene], scene_collision_weight, activation_distance, batch_size: int, dof: int,
    env_query_idx=None, num_envs: int = 1, use_multi_env: bool = False,
):
    """cost[b], d cost / d q [b, dof] of ``batch_size`` joint configurations against per-row goal
    poses in one launch (see ``curobo_hip_rollout_ik_fused`` in the header); optional outputs may be None.
    ``env_query_idx`` [b] (scene environment with ``use_multi_env``, sphere set with ``num_envs`` > 1) must be constant
    over aligned runs of 16 configurations."""
    num_pairs = 0 if pair_locations is None else int(pair_locations.shape[0])
    check(load().curobo_hip_rollout_ik_fused(
        ptr(out_cost), ptr(out_grad_q), ptr(out_pose_distance), ptr(out_position_distance), ptr(out_rotation_distance),
        ptr(out_goalset_idx), ptr(out_link_pos), ptr(out_link_quat), ptr(out_robot_spheres), ptr(out_cspace_cost),
        ptr(q), ptr(goal_position), ptr(goal_quat), ptr(idxs_goal), ptr(position_orientation_weight),
        ptr(terminal_pose_axes_weight_factor), ptr(terminal_pose_convergence_tolerance), ptr(project_distance_to_goal),
        num_goalset, rotation_method, ptr(p_b), ptr(cspace_weight), ptr(cspace_activation_distance), ptr(fixed_transform),
        ptr(robot_spheres), ptr(joint_map_type), ptr(joint_map), ptr(link_map), ptr(tool_frame_map), ptr(link_sphere_map),
        ptr(link_chain_data), ptr(link_chain_offsets), ptr(joint_offset_map), ptr(sphere_padding),
        ptr(self_collision_weight), ptr(pair_locations), None if scene is None else C.addressof(scene),
        ptr(scene_collision_weight), ptr(activation_distance), batch_size, dof, int(fixed_transform.shape[0]),
        int(tool_frame_map.shape[0]), int(link_sphere_map.shape[0]), num_pairs, int(link_chain_data.shape[0]),
        ptr(env_query_idx), int(num_envs), int(use_multi_env), current_stream(out_cost),
    ))


def rollout_ik_fused_lds_bytes(dof: int, num_links: int, num_spheres: int, num_collision_pairs: int,
                               link_chain_len: int, num_obstacles: int) -> int:
    return int(load().curobo_hip_rollout_ik_fused_lds_bytes(dof, num_links, num_spheres, num_collision_pairs,
                                                            link_chain_len, num_obstacles))


def make_trajopt_terms(**tensors) -> TrajOptTerms:
    """Pack device tensors / ints into ``curobo_hip_trajopt_terms`` (field names of the header; the
    caller keeps the tensors alive).  Omitted fields stay NULL / 0."""
    t = TrajOptTerms()
    for name, value in tensors.items():
        setattr(t, name, int(value) if isinstance(value, (int, bool)) else ptr(value))
    return t


def rollout_trajopt_fused(terms: Optional[TrajOptTerms], *args, **kwargs):
    """``rollout_trajectory_fused`` plus the optional tool-pose and c-space STATE terms of the
    reference trajopt task in the same launch (``curobo_hip_rollout_trajopt_fused``).  Positional
    arguments after ``terms`` are those of :func:`rollout_trajectory_fused`."""
    return _rollout_trajectory(load().curobo_hip_rollout_trajopt_fused, terms, True, *args, **kwargs)


def rollout_trajopt_fused_torque_fits(padded_horizon: int, dof: int, num_links: int, num_spheres: int,
                                      num_collision_pairs: int, link_chain_len: int, num_obstacles: int) -> bool:
    """whether the torque-limit terms (RNEA forward / VJP inside the launch) fit the LDS regions they borrow"""
    return bool(load().curobo_hip_rollout_trajopt_fused_torque_fits(
        padded_horizon, dof, num_links, num_spheres, num_collision_pairs, link_chain_len, num_obstacles))


def rollout_trajopt_fused_lds_bytes(padded_horizon: int, dof: int, num_links: int, num_spheres: int,
                                    num_collision_pairs: int, link_chain_len: int, num_obstacles: int,
                                    with_cspace_terms: bool) -> int:
    return int(load().curobo_hip_rollout_trajopt_fused_lds_bytes(
        padded_horizon, dof, num_links, num_spheres, num_collision_pairs, link_chain_len, num_obstacles,
        int(with_cspace_terms)))


def fused_shape_id(padded_horizon: int, n_knots: int, dof: int, num_links: int, num_spheres: int, num_collision_pairs: int,
                   link_chain_len: int, self_lane_len: int, max_cuboids: int, max_voxel_grids: int, bspline_degree: int = 3,
                   sweep_steps: int = 3, kinds: int = 1, with_trajopt_terms: bool = False, plain_launch: bool = True) -> int:
    """Which compile-time shape (csrc/fused_shapes.hpp) a fused trajectory launch with these dimensions runs; 0 = the generic
    kernel.  ``plain_launch``: the launch form of an optimiser iteration (self + scene collision with the speed metric, one
    environment, longest-first dispatch, nothing materialised).  Host-side query, no GPU work."""
    return int(load().curobo_hip_rollout_fused_shape_id(
        int(padded_horizon), int(n_knots), int(dof), int(num_links), int(num_spheres), int(num_collision_pairs), int(link_chain_len),
        int(self_lane_len), int(max_cuboids), int(max_voxel_grids), int(bspline_degree), int(sweep_steps), int(kinds),
        1 if with_trajopt_terms else 0, 1 if plain_launch else 0))


def set_fused_shapes_enabled(enabled: bool) -> None:
    """False: every fused trajectory launch takes the generic kernel (same results; A/B timing and the bit-equality tests)."""
    check(load().curobo_hip_rollout_fused_set_shapes_enabled(1 if enabled else 0))
