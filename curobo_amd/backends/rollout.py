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

from .._lib import Scene, check, current_stream, load, ptr


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
):
    """cost[b], grad_knots[b, n_knots, dof] of ``batch_size`` B-spline trajectories in one launch."""
    num_pairs = 0 if pair_locations is None else int(pair_locations.shape[0])
    check(load().curobo_hip_rollout_trajectory_fused(
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
        int(enable_speed_metric), current_stream(out_cost),
    ))


FUSED_LDS_LIMIT = 160 * 1024


def rollout_trajectory_fused_lds_bytes(padded_horizon: int, dof: int, num_links: int, num_spheres: int,
                                       num_collision_pairs: int, link_chain_len: int, num_obstacles: int) -> int:
    """LDS bytes one trajectory needs in the fused kernel (usable when <= FUSED_LDS_LIMIT)."""
    return int(load().curobo_hip_rollout_trajectory_fused_lds_bytes(
        padded_horizon, dof, num_links, num_spheres, num_collision_pairs, link_chain_len, num_obstacles))
