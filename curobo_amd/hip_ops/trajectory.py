"""``BSplineIdxKernel`` -- reference ``cuda_ops/trajectory.py:299-456``."""

from __future__ import annotations

import torch

from ..backends import trajectory as trajectory_hip
from .tensor_checks import check_float32_tensors, check_int32_tensors, check_uint8_tensors


class BSplineIdxKernel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u_act, start_position, start_velocity, start_acceleration, start_jerk, goal_position,
                goal_velocity, goal_acceleration, goal_jerk, start_idx, goal_idx, out_position, out_velocity,
                out_acceleration, out_jerk, out_dt, traj_dt, use_implicit_goal_state, out_grad_position,
                bspline_degree, use_flat_gradient=False):
        n_knots = u_act.shape[-2]
        device = u_act.device
        check_float32_tensors(
            device, u_act=u_act, start_position=start_position, start_velocity=start_velocity,
            start_acceleration=start_acceleration, start_jerk=start_jerk, goal_position=goal_position,
            goal_velocity=goal_velocity, goal_acceleration=goal_acceleration, goal_jerk=goal_jerk,
            out_position=out_position, out_velocity=out_velocity, out_acceleration=out_acceleration,
            out_jerk=out_jerk, out_dt=out_dt, traj_dt=traj_dt, out_grad_position=out_grad_position)
        check_int32_tensors(device, start_idx=start_idx, goal_idx=goal_idx)
        check_uint8_tensors(device, use_implicit_goal_state=use_implicit_goal_state)
        trajectory_hip.launch_bspline_interpolation_forward_kernel(
            out_position, out_velocity, out_acceleration, out_jerk, out_dt, u_act, start_position,
            start_velocity, start_acceleration, start_jerk, goal_position, goal_velocity, goal_acceleration,
            goal_jerk, start_idx, goal_idx, traj_dt, use_implicit_goal_state, out_position.shape[0],
            out_position.shape[1], out_position.shape[-1], n_knots, bspline_degree)
        ctx.use_flat_gradient = use_flat_gradient
        ctx.save_for_backward(traj_dt, out_grad_position, goal_idx, use_implicit_goal_state)
        ctx.n_knots, ctx.bspline_degree = n_knots, bspline_degree
        return out_position, out_velocity, out_acceleration, out_jerk

    @staticmethod
    def backward(ctx, grad_out_p, grad_out_v, grad_out_a, grad_out_j):
        u_grad = None
        if ctx.needs_input_grad[0]:
            traj_dt, out_grad_position, dt_idx, use_implicit_goal_state = ctx.saved_tensors
            padded_horizon = grad_out_p.shape[1]
            zeros = None
            grads = []
            for g in (grad_out_p, grad_out_v, grad_out_a, grad_out_j):
                if g is None:  # unused output: contribute zeros
                    if zeros is None:
                        zeros = torch.zeros_like(out_grad_position.new_empty(
                            out_grad_position.shape[0], padded_horizon, out_grad_position.shape[-1]))
                    g = zeros
                grads.append(g.contiguous())
            check_float32_tensors(grads[0].device, out_grad_position=out_grad_position, traj_dt=traj_dt)
            trajectory_hip.launch_bspline_interpolation_backward_kernel(
                out_grad_position, grads[0], grads[1], grads[2], grads[3], traj_dt, dt_idx,
                use_implicit_goal_state, grads[0].shape[0], grads[0].shape[1], grads[0].shape[2], ctx.n_knots,
                ctx.bspline_degree, ctx.use_flat_gradient)
            u_grad = out_grad_position
        return (u_grad,) + (None,) * 20
