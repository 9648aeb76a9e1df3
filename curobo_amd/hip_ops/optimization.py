"""``LBFGScu`` and ``wolfe_line_search`` -- reference ``cuda_ops/optimization.py:22-251``."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from ..backends import optimization as optimization_hip
from .tensor_checks import (
    check_float32_tensors,
    check_int16_tensors,
    check_int32_tensors,
    check_uint8_tensors,
)


@dataclass
class OptimizationIterationState:
    """The tensors the line-search kernel updates (reference
    ``optim/components/optimization_iteration_state.py``): shapes (num_problems,) and
    (num_problems, action_horizon, action_dim)."""

    action: torch.Tensor
    gradient: torch.Tensor
    cost: torch.Tensor
    best_action: torch.Tensor
    best_cost: torch.Tensor
    best_iteration: torch.Tensor
    current_iteration: torch.Tensor
    converged: torch.Tensor
    exploration_action: torch.Tensor
    exploration_gradient: torch.Tensor
    exploration_cost: torch.Tensor
    step_direction: Optional[torch.Tensor] = None


@dataclass
class LineSearchContext:
    """Subset of the reference ``LineSearchContext`` (optim/gradient/line_search_context.py:12-48)
    the kernel path reads."""

    num_problems: int
    opt_dim: int
    action_horizon: int
    action_dim: int
    line_search_scale: torch.Tensor  # (1, n_linesearch, 1, 1)
    line_search_c_1: float
    line_search_c_2: float
    convergence_iteration: int
    cost_delta_threshold: float
    cost_relative_threshold: float

    @property
    def n_linesearch(self) -> int:
        return self.line_search_scale.shape[1]


def wolfe_line_search(iteration_state: OptimizationIterationState, line_search_context: LineSearchContext,
                      exploration_idx, selected_idx, search_cost, search_action, search_gradient, step_direction,
                      strong_wolfe: bool, approx_wolfe: bool):
    """reference :22-189 (same shape checks, same in-place update of ``iteration_state``)."""
    c = line_search_context
    n, v, nls = c.num_problems, c.opt_dim, c.n_linesearch
    s = iteration_state
    device = s.best_cost.device
    check_float32_tensors(
        device, best_cost=s.best_cost, best_action=s.best_action, exploration_cost=s.exploration_cost,
        exploration_action=s.exploration_action, exploration_gradient=s.exploration_gradient, cost=s.cost,
        action=s.action, gradient=s.gradient, search_cost=search_cost, search_action=search_action,
        search_gradient=search_gradient, step_direction=step_direction, line_search_scale=c.line_search_scale)
    check_int16_tensors(device, best_iteration=s.best_iteration, current_iteration=s.current_iteration)
    check_uint8_tensors(device, converged=s.converged)
    check_int32_tensors(device, exploration_idx=exploration_idx, selected_idx=selected_idx)
    hd = (n, c.action_horizon, c.action_dim)
    expect = {
        "best_cost": (s.best_cost, (n,)), "best_action": (s.best_action, hd),
        "best_iteration": (s.best_iteration, (n,)), "current_iteration": (s.current_iteration, (n,)),
        "converged": (s.converged, (n,)), "exploration_cost": (s.exploration_cost, (n,)),
        "exploration_action": (s.exploration_action, hd), "exploration_gradient": (s.exploration_gradient, hd),
        "cost": (s.cost, (n,)), "action": (s.action, hd), "gradient": (s.gradient, hd),
        "exploration_idx": (exploration_idx, (n, nls)), "selected_idx": (selected_idx, (n, nls)),
        "search_cost": (search_cost, (n, nls, 1)), "search_action": (search_action, (n, nls, v)),
        "search_gradient": (search_gradient, (n, nls, v)), "step_direction": (step_direction, (n, 1, v)),
        "line_search_scale": (c.line_search_scale, (1, nls, 1, 1)),
    }
    for name, (t, shape) in expect.items():
        if tuple(t.shape) != shape:
            raise ValueError(f"{name} must have shape {shape}. Got {tuple(t.shape)}")
    optimization_hip.launch_line_search(
        s.best_cost, s.best_action, s.best_iteration, s.current_iteration, s.converged, c.convergence_iteration,
        c.cost_delta_threshold, c.cost_relative_threshold, s.exploration_cost, s.exploration_action,
        s.exploration_gradient, exploration_idx.view(-1), s.cost, s.action, s.gradient, selected_idx.view(-1),
        search_cost, search_action, search_gradient, step_direction, c.line_search_scale, c.line_search_c_1,
        c.line_search_c_2, strong_wolfe, approx_wolfe, nls, v, n)
    return iteration_state, exploration_idx, selected_idx


class LBFGScu(torch.autograd.Function):
    """reference :192-251: ``apply(step_vec, rho_buffer, y_buffer, s_buffer, q, grad_q, x_0, grad_0,
    epsilon, stable_mode, use_shared_buffers) -> step_vec``; buffers shaped (m, B, V, 1) /
    (m, B, 1, 1) / (B, V, 1) like ``QuasiNewtonBuffers``."""

    @staticmethod
    def forward(ctx, step_vec, rho_buffer, y_buffer, s_buffer, q, grad_q, x_0, grad_0, epsilon: float = 0.1,
                stable_mode: bool = False, use_shared_buffers: bool = True):
        device = step_vec.device
        check_float32_tensors(device, step_vec=step_vec, rho_buffer=rho_buffer, y_buffer=y_buffer,
                              s_buffer=s_buffer, q=q, grad_q=grad_q, x_0=x_0, grad_0=grad_0)
        m, b, v_dim = y_buffer.shape[0], y_buffer.shape[1], y_buffer.shape[2]
        optimization_hip.launch_lbfgs_step(step_vec, rho_buffer, y_buffer, s_buffer, q, grad_q, x_0, grad_0,
                                           epsilon, b, m, v_dim, stable_mode, use_shared_buffers)
        return step_vec

    @staticmethod
    def backward(ctx, grad_output):
        return (None,) * 11
