"""``torch.autograd.Function`` wrappers of the cost kernels with the reference's class names."""

from __future__ import annotations

import torch

from ..backends import cost as cost_hip
from .tensor_checks import check_float32_tensors


class L2DistFunction(torch.autograd.Function):
    """reference ``L2DistFunction`` (``curobo/_src/cost/wp_torch_cspace_dist.py:81-158``): same
    argument list; returns ``cost[batch, horizon]``, gradient to ``pos`` from the buffer the forward wrote."""

    @staticmethod
    def forward(ctx, pos, target, target_idx, weight, terminal_dof_weight, non_terminal_dof_weight, out_cost_dof,
                out_gp, use_grad_input: bool):
        b, h, dof = pos.shape
        p = pos.detach().contiguous()
        check_float32_tensors(p.device, pos=p, target=target, weight=weight)
        cost_hip.cspace_l2_distance(out_cost_dof, out_gp, p, target, target_idx, weight, terminal_dof_weight,
                                    non_terminal_dof_weight, pos.requires_grad, b, h, dof)
        ctx.save_for_backward(out_gp)
        ctx.use_grad_input = use_grad_input
        return torch.sum(out_cost_dof, dim=-1)

    @staticmethod
    def backward(ctx, grad_out_cost):
        (p_grad,) = ctx.saved_tensors
        p_g = None
        if ctx.needs_input_grad[0]:
            p_g = p_grad * grad_out_cost.unsqueeze(-1) if ctx.use_grad_input else p_grad
        return p_g, None, None, None, None, None, None, None, None


class ToolPoseDistance(torch.autograd.Function):
    """reference ``ToolPoseDistance`` (``curobo/_src/cost/wp_tool_pose.py:696-914``): same argument
    order without the Warp kernel handle (``num_goalset`` / ``rotation_method`` are plain arguments
    here); gradients flow to ``current_position`` / ``current_quat`` only, from the buffers the
    forward launch wrote; ``out_distance`` interleaves (position cost, rotation cost) per link."""

    @staticmethod
    def forward(ctx, current_position, current_quat, goal_position, goal_quat, idxs_goal, position_orientation_weight,
                terminal_pose_axes_weight_factor, non_terminal_pose_axes_weight_factor,
                terminal_pose_convergence_tolerance, non_terminal_pose_convergence_tolerance, project_distance_to_goal,
                out_distance, out_position_distance, out_rotation_distance, out_position_gradient, out_rotation_gradient,
                out_goalset_idx, use_grad_input: bool, num_goalset: int = 1, rotation_method: int = 0):
        b, h, num_links, _ = current_position.shape
        cp, cq = current_position.detach().contiguous(), current_quat.detach().contiguous()
        check_float32_tensors(cp.device, current_position=cp, current_quat=cq, goal_position=goal_position, goal_quat=goal_quat)
        cost_hip.tool_pose_distance(
            out_distance, out_position_distance, out_rotation_distance, out_position_gradient, out_rotation_gradient,
            out_goalset_idx, cp, cq, goal_position, goal_quat, idxs_goal, position_orientation_weight,
            terminal_pose_axes_weight_factor, non_terminal_pose_axes_weight_factor, terminal_pose_convergence_tolerance,
            non_terminal_pose_convergence_tolerance, project_distance_to_goal, b, h, num_links, num_goalset, rotation_method)
        ctx.use_grad_input = use_grad_input
        ctx.mark_non_differentiable(out_position_distance, out_rotation_distance, out_goalset_idx)
        ctx.save_for_backward(out_position_gradient, out_rotation_gradient)
        return out_distance, out_position_distance, out_rotation_distance, out_goalset_idx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_distance, grad_position_distance, grad_rotation_distance, grad_goalset_idx):
        pos_grad = quat_grad = None
        if grad_distance is not None:
            gp, gr = ctx.saved_tensors
            if ctx.needs_input_grad[0]:
                pos_grad = gp * grad_distance[:, :, 0::2].unsqueeze(-1) if ctx.use_grad_input else gp
            if ctx.needs_input_grad[1]:
                quat_grad = gr * grad_distance[:, :, 1::2].unsqueeze(-1) if ctx.use_grad_input else gr
        return (pos_grad, quat_grad) + (None,) * 18


class StateCSpaceFunction(torch.autograd.Function):
    """reference ``StateCSpaceFunction`` (``curobo/_src/cost/wp_cspace_state.py:288-680``), same argument
    order: joint-limit / velocity / acceleration / jerk / effort bounds, squared-L2 regularisation and
    the optional configuration target in one launch; returns ``out_cost[B,H,D]``; gradients to
    position, velocity, acceleration, jerk and effort from the buffers written by the forward launch."""

    @staticmethod
    def forward(ctx, pos, vel, acc, jerk, effort, state_dt, target_joint_position, idxs_target_joint_position, p_b, v_b,
                a_b, j_b, effort_b, weight, activation_distance, squared_l2_regularization_weight, cspace_target_weight,
                cspace_non_terminal_weight_factor, cspace_target_dof_weight, out_cost, out_gp, out_gv, out_ga, out_gj,
                out_gtau, retime_weights: bool, retime_regularization_weights: bool, use_grad_input: bool):
        b, h, dof = pos.shape
        det = lambda t: None if t is None else t.detach().contiguous()  # noqa: E731
        cost_hip.cspace_state_cost(
            out_cost, out_gp, out_gv, out_ga, out_gj, out_gtau, det(pos), det(vel), det(acc), det(jerk), det(effort), state_dt,
            target_joint_position, idxs_target_joint_position, p_b, v_b, a_b, j_b, effort_b, weight, activation_distance,
            squared_l2_regularization_weight, cspace_target_weight, cspace_non_terminal_weight_factor,
            cspace_target_dof_weight, True, b, h, dof, retime_weights, retime_regularization_weights)
        ctx.use_grad_input = use_grad_input
        ctx.has_effort = effort is not None
        ctx.save_for_backward(out_gp, out_gv, out_ga, out_gj, out_gtau)
        return out_cost

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out_cost):
        grads = [None] * 5
        if grad_out_cost is not None:
            for i, g in enumerate(ctx.saved_tensors):
                if ctx.needs_input_grad[i] and (i < 4 or ctx.has_effort):
                    grads[i] = g * grad_out_cost if ctx.use_grad_input else g
        return tuple(grads) + (None,) * 23
