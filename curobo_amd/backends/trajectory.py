"""HIP backend for the B-spline trajectory kernels (reference
``curobo/_src/curobolib/backends/cuda_core_backend/trajectory.py:28-204``,
``pybind/trajectory_bindings.cpp:133-142``)."""

from __future__ import annotations

import torch

from .._lib import check, current_stream, load, ptr


def launch_bspline_interpolation_forward_kernel(
    out_position: torch.Tensor,
    out_velocity: torch.Tensor,
    out_acceleration: torch.Tensor,
    out_jerk: torch.Tensor,
    out_dt: torch.Tensor,
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
    batch_size: int,
    horizon: int,
    dof: int,
    n_knots: int,
    bspline_degree: int,
):
    check(load().curobo_hip_launch_bspline_interpolation_forward_kernel(
        ptr(out_position), ptr(out_velocity), ptr(out_acceleration), ptr(out_jerk), ptr(out_dt),
        ptr(u_position), ptr(start_position), ptr(start_velocity), ptr(start_acceleration),
        ptr(start_jerk), ptr(goal_position), ptr(goal_velocity), ptr(goal_acceleration),
        ptr(goal_jerk), ptr(start_idx), ptr(goal_idx), ptr(traj_dt), ptr(use_implicit_goal_state),
        batch_size, horizon, dof, n_knots, bspline_degree, current_stream(out_position),
    ))


def launch_bspline_interpolation_backward_kernel(
    out_grad_position: torch.Tensor,
    grad_position: torch.Tensor,
    grad_velocity: torch.Tensor,
    grad_acceleration: torch.Tensor,
    grad_jerk: torch.Tensor,
    traj_dt: torch.Tensor,
    dt_idx: torch.Tensor,
    use_implicit_goal_state: torch.Tensor,
    batch_size: int,
    padded_horizon: int,
    dof: int,
    n_knots: int,
    bspline_degree: int,
    use_direct_polynomial: bool,
):
    if padded_horizon - 1 < 5:
        raise RuntimeError("horizon must be greater than 5")  # reference :157-158
    check(load().curobo_hip_launch_bspline_interpolation_backward_kernel(
        ptr(out_grad_position), ptr(grad_position), ptr(grad_velocity), ptr(grad_acceleration),
        ptr(grad_jerk), ptr(traj_dt), ptr(dt_idx), ptr(use_implicit_goal_state), batch_size,
        padded_horizon, dof, n_knots, bspline_degree, int(use_direct_polynomial),
        current_stream(out_grad_position),
    ))


def launch_bspline_interpolation_single_dt_kernel(
    out_position: torch.Tensor,
    out_velocity: torch.Tensor,
    out_acceleration: torch.Tensor,
    out_jerk: torch.Tensor,
    out_dt: torch.Tensor,
    knots: torch.Tensor,
    knot_dt: torch.Tensor,
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
    interpolation_dt: torch.Tensor,
    use_implicit_goal_state: torch.Tensor,
    interpolation_horizon: torch.Tensor,
    batch_size: int,
    max_out_tsteps: int,
    dof: int,
    n_knots: int,
    bspline_degree: int,
):
    """reference cuda_core_backend/trajectory.py:207-306 (re-interpolation of solved knots at one
    dt with a per-trajectory horizon; ``knot_dt`` is unused there as well)."""
    check(load().curobo_hip_launch_bspline_interpolation_single_dt_kernel(
        ptr(out_position), ptr(out_velocity), ptr(out_acceleration), ptr(out_jerk), ptr(out_dt), ptr(knots),
        ptr(knot_dt), ptr(start_position), ptr(start_velocity), ptr(start_acceleration), ptr(start_jerk),
        ptr(goal_position), ptr(goal_velocity), ptr(goal_acceleration), ptr(goal_jerk), ptr(start_idx),
        ptr(goal_idx), ptr(interpolation_dt), ptr(use_implicit_goal_state), ptr(interpolation_horizon),
        batch_size, max_out_tsteps, dof, n_knots, bspline_degree, current_stream(out_position),
    ))


def launch_differentiation_position_forward_kernel(
    out_position: torch.Tensor,
    out_velocity: torch.Tensor,
    out_acceleration: torch.Tensor,
    out_jerk: torch.Tensor,
    out_dt: torch.Tensor,
    u_position: torch.Tensor,
    start_position: torch.Tensor,
    start_velocity: torch.Tensor,
    start_acceleration: torch.Tensor,
    goal_position: torch.Tensor,
    goal_velocity: torch.Tensor,
    goal_acceleration: torch.Tensor,
    start_idx: torch.Tensor,
    goal_idx: torch.Tensor,
    traj_dt: torch.Tensor,
    use_implicit_goal_state: torch.Tensor,
    batch_size: int,
    horizon: int,
    dof: int,
):
    """POSITION control space: reference cuda_core_backend/trajectory.py:309-393."""
    check(load().curobo_hip_launch_differentiation_position_forward_kernel(
        ptr(out_position), ptr(out_velocity), ptr(out_acceleration), ptr(out_jerk), ptr(out_dt), ptr(u_position),
        ptr(start_position), ptr(start_velocity), ptr(start_acceleration), ptr(goal_position), ptr(goal_velocity),
        ptr(goal_acceleration), ptr(start_idx), ptr(goal_idx), ptr(traj_dt), ptr(use_implicit_goal_state),
        batch_size, horizon, dof, current_stream(out_position),
    ))


def launch_differentiation_position_backward_kernel(
    out_grad_position: torch.Tensor,
    grad_position: torch.Tensor,
    grad_velocity: torch.Tensor,
    grad_acceleration: torch.Tensor,
    grad_jerk: torch.Tensor,
    traj_dt: torch.Tensor,
    dt_idx: torch.Tensor,
    use_implicit_goal_state: torch.Tensor,
    batch_size: int,
    horizon: int,
    dof: int,
):
    """reference cuda_core_backend/trajectory.py:396-465."""
    check(load().curobo_hip_launch_differentiation_position_backward_kernel(
        ptr(out_grad_position), ptr(grad_position), ptr(grad_velocity), ptr(grad_acceleration), ptr(grad_jerk),
        ptr(traj_dt), ptr(dt_idx), ptr(use_implicit_goal_state), batch_size, horizon, dof,
        current_stream(out_grad_position),
    ))


def launch_integration_acceleration_kernel(
    out_position: torch.Tensor,
    out_velocity: torch.Tensor,
    out_acceleration: torch.Tensor,
    out_jerk: torch.Tensor,
    u_acc: torch.Tensor,
    start_position: torch.Tensor,
    start_velocity: torch.Tensor,
    start_acceleration: torch.Tensor,
    start_idx: torch.Tensor,
    traj_dt: torch.Tensor,
    batch_size: int,
    horizon: int,
    dof: int,
    use_rk2: bool = True,
):
    """ACCELERATION control space: reference cuda_core_backend/trajectory.py:468-556."""
    check(load().curobo_hip_launch_integration_acceleration_kernel(
        ptr(out_position), ptr(out_velocity), ptr(out_acceleration), ptr(out_jerk), ptr(u_acc),
        ptr(start_position), ptr(start_velocity), ptr(start_acceleration), ptr(start_idx), ptr(traj_dt),
        batch_size, horizon, dof, int(use_rk2), current_stream(out_position),
    ))
