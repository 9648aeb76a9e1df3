"""Trajectory retiming and re-interpolation of solved B-spline knots.

Mirrors ``curobo/_src/util/trajectory.py``: ``calculate_dt_no_clamp`` (:235-259, the time scale at
which the worst joint just meets its velocity / acceleration / jerk limit), ``calculate_traj_steps``
(:262-280, samples per trajectory at the interpolation dt) and the ``BSPLINE_KNOTS_CUDA`` branch of
``get_batch_interpolated_trajectory`` (:39-141) on the single-dt B-spline kernel
(``curobo_hip_launch_bspline_interpolation_single_dt_kernel``).
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch


def calculate_dt_no_clamp(vel: torch.Tensor, acc: torch.Tensor, jerk: torch.Tensor, max_vel: torch.Tensor,
                          max_acc: torch.Tensor, max_jerk: torch.Tensor, epsilon: float = 1e-5) -> torch.Tensor:
    """[..., H, D] derivatives -> [...] factor by which dt must be scaled so that every joint stays
    within its limits: velocity scales with 1/s, acceleration with 1/s^2, jerk with 1/s^3."""
    d = vel.shape[-1]
    sv = (vel.abs().amax(dim=-2) / max_vel.reshape(1, d)).amax(dim=-1)
    sa = (acc.abs().amax(dim=-2) / max_acc.reshape(1, d)).amax(dim=-1).pow(1.0 / 2.0)
    sj = (jerk.abs().amax(dim=-2) / max_jerk.reshape(1, d)).amax(dim=-1).pow(1.0 / 3.0)
    return torch.maximum(torch.maximum(sv, sa), sj) * (1.0 + epsilon)


def calculate_traj_steps(opt_dt: torch.Tensor, interpolation_dt: torch.Tensor, horizon: int,
                         nearest_int: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(samples per trajectory [B] int32, their maximum) when a trajectory of ``horizon`` knots /
    waypoints spaced ``opt_dt`` is resampled at ``interpolation_dt``"""
    if nearest_int:
        per = (opt_dt + interpolation_dt) / interpolation_dt
    else:
        per = (opt_dt + opt_dt % interpolation_dt) / interpolation_dt
    steps = (horizon - 1) * per.to(torch.int64) + 1
    return steps.to(torch.int32), steps.max().to(torch.int32)


def interpolate_bspline_knots(knots: torch.Tensor, knot_dt: torch.Tensor, interpolation_dt: float, start_state,
                              goal_state=None, use_implicit_goal_state: Optional[torch.Tensor] = None,
                              bspline_degree: int = 3, out_steps: Optional[int] = None, out_steps_is_bound: bool = False):
    """knots [B, n_knots, D] with knot spacing ``knot_dt`` [B] -> (position, velocity, acceleration,
    jerk) [B, steps_max, D] sampled every ``interpolation_dt`` and ``last_tstep`` [B] (samples beyond
    it repeat the final state).  ``start_state`` / ``goal_state`` = tuples of (position, velocity,
    acceleration, jerk) [1 or B, D]; a missing goal means "rest at the last knot"."""
    from ..backends import trajectory as trajectory_hip

    B, n_knots, D = knots.shape
    dev = knots.device
    total = n_knots + bspline_degree + 1  # reference ControlSpace.spline_total_knots
    idt = torch.full((B,), float(interpolation_dt), device=dev)
    steps, steps_max = calculate_traj_steps(knot_dt.to(dev, torch.float32), idt, total + 1, nearest_int=True)
    if out_steps_is_bound:
        # the caller's buffer holds the longest trajectory its dt range allows: nothing is read back (graph capture); a
        # trajectory that claims more samples is cut at the buffer
        n_out = int(out_steps)
        steps = torch.clamp(steps, max=n_out)
    else:
        n_out = int(out_steps) if out_steps is not None else int(steps_max)
        if n_out < int(steps_max):
            raise ValueError(f"interpolation buffer ({n_out} steps) is smaller than the trajectory ({int(steps_max)} steps)")
    z = lambda: torch.zeros(B, n_out, D, device=dev)  # noqa: E731
    out = [z(), z(), z(), z()]
    out_dt = torch.zeros(B, device=dev)
    s = [t.to(dev, torch.float32).reshape(-1, D).contiguous() for t in start_state]
    g = s if goal_state is None else [t.to(dev, torch.float32).reshape(-1, D).contiguous() for t in goal_state]
    sidx = torch.zeros(B, dtype=torch.int32, device=dev) if s[0].shape[0] == 1 else torch.arange(B, dtype=torch.int32, device=dev)
    gidx = torch.zeros(B, dtype=torch.int32, device=dev) if g[0].shape[0] == 1 else torch.arange(B, dtype=torch.int32, device=dev)
    implicit = use_implicit_goal_state if use_implicit_goal_state is not None else torch.zeros(g[0].shape[0], dtype=torch.uint8, device=dev)
    trajectory_hip.launch_bspline_interpolation_single_dt_kernel(
        *out, out_dt, knots.contiguous(), knot_dt.to(dev, torch.float32).contiguous(), *s, *g, sidx, gidx, idt, implicit,
        steps.contiguous(), B, n_out, D, n_knots, bspline_degree)
    return out, steps
