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
