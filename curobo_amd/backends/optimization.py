"""HIP backend for the optimiser-step kernels (reference
``curobo/_src/curobolib/backends/cuda_core_backend/optimization.py:27-260``,
``pybind/optimization_bindings.cpp:14-75``)."""

from __future__ import annotations

from typing import List

import torch

from .._lib import check, current_stream, load, ptr


def launch_line_search(
    best_cost: torch.Tensor,
    best_action: torch.Tensor,
    best_iteration: torch.Tensor,
    current_iteration: torch.Tensor,
    converged_global: torch.Tensor,
    convergence_iteration: int,
    cost_delta_threshold: float,
    cost_relative_threshold: float,
    exploration_cost: torch.Tensor,
    exploration_action: torch.Tensor,
    exploration_gradient: torch.Tensor,
    exploration_idx: torch.Tensor,
    selected_cost: torch.Tensor,
    selected_action: torch.Tensor,
    selected_gradient: torch.Tensor,
    selected_idx: torch.Tensor,
    search_cost: torch.Tensor,
    search_action: torch.Tensor,
    search_gradient: torch.Tensor,
    step_direction: torch.Tensor,
    search_magnitudes: torch.Tensor,
    armijo_threshold_c_1: float,
    curvature_threshold_c_2: float,
    strong_wolfe: bool,
    approx_wolfe: bool,
    n_linesearch: int,
    opt_dim: int,
    batchsize: int,
):
    check(load().curobo_hip_launch_line_search(
        ptr(best_cost), ptr(best_action), ptr(best_iteration), ptr(current_iteration),
        ptr(converged_global), convergence_iteration, cost_delta_threshold,
        cost_relative_threshold, ptr(exploration_cost), ptr(exploration_action),
        ptr(exploration_gradient), ptr(exploration_idx), ptr(selected_cost), ptr(selected_action),
        ptr(selected_gradient), ptr(selected_idx), ptr(search_cost), ptr(search_action),
        ptr(search_gradient), ptr(step_direction), ptr(search_magnitudes), armijo_threshold_c_1,
        curvature_threshold_c_2, int(strong_wolfe), int(approx_wolfe), n_linesearch, opt_dim,
        batchsize, current_stream(best_cost),
    ))


def launch_lbfgs_step(
    step_vec: torch.Tensor,
    rho_buffer: torch.Tensor,
    y_buffer: torch.Tensor,
    s_buffer: torch.Tensor,
    q: torch.Tensor,
    grad_q: torch.Tensor,
    x_0: torch.Tensor,
    grad_0: torch.Tensor,
    epsilon: float,
    batch_size: int,
    history_m: int,
    v_dim: int,
    stable_mode: bool,
    use_shared_buffers: bool,
) -> List[torch.Tensor]:
    """Returns ``[step_vec, rho_buffer, y_buffer, s_buffer, x_0, grad_0]`` like the reference."""
    check(load().curobo_hip_launch_lbfgs_step(
        ptr(step_vec), ptr(rho_buffer), ptr(y_buffer), ptr(s_buffer), ptr(q), ptr(grad_q),
        ptr(x_0), ptr(grad_0), epsilon, batch_size, history_m, v_dim, int(stable_mode),
        int(use_shared_buffers), current_stream(step_vec),
    ))
    return [step_vec, rho_buffer, y_buffer, s_buffer, x_0, grad_0]


def prepare_search_points(
    x_set: torch.Tensor,
    step_direction_out: torch.Tensor,
    x: torch.Tensor,
    step_direction: torch.Tensor,
    action_step_max: torch.Tensor,
    search_magnitudes: torch.Tensor,
    batchsize: int,
    n_linesearch: int,
    opt_dim: int,
    action_dim: int,
    apply_step_scale: bool,
):
    """Fused ``_prepare_search_points`` (reference optim/gradient/line_search_strategy.py:134-204):
    step clamping + ``x + alpha_k * d`` for every line-search magnitude, one wavefront per problem."""
    check(load().curobo_hip_prepare_search_points(
        ptr(x_set), ptr(step_direction_out), ptr(x), ptr(step_direction), ptr(action_step_max),
        ptr(search_magnitudes), batchsize, n_linesearch, opt_dim, action_dim, int(apply_step_scale),
        current_stream(x_set),
    ))


def launch_lbfgs_iteration_tail(
    best_cost, best_action, best_iteration, current_iteration, converged_global, convergence_iteration: int,
    cost_delta_threshold: float, cost_relative_threshold: float, exploration_cost, exploration_action,
    exploration_gradient, exploration_idx, selected_cost, selected_action, selected_gradient, selected_idx,
    search_cost, search_action, search_gradient, step_direction_scaled, search_magnitudes,
    armijo_threshold_c_1: float, curvature_threshold_c_2: float, strong_wolfe: bool, approx_wolfe: bool,
    n_linesearch: int, opt_dim: int, batchsize: int, step_vec, rho_buffer, y_buffer, s_buffer, x_0, grad_0,
    epsilon: float, history_m: int, stable_mode: bool, action_step_max, action_dim: int, apply_step_scale: bool,
    overlapped: bool = False,
):
    """``launch_line_search`` + ``launch_lbfgs_step`` + ``prepare_search_points`` (next iteration) in
    one launch; argument groups in that order, same meaning as in the three functions above.
    ``overlapped``: the launch runs next to other kernels (seed shards on their own streams): one wavefront and no
    LDS per problem instead of a workgroup (see the header)."""
    check(load().curobo_hip_launch_lbfgs_iteration_tail(
        ptr(best_cost), ptr(best_action), ptr(best_iteration), ptr(current_iteration), ptr(converged_global),
        convergence_iteration, cost_delta_threshold, cost_relative_threshold, ptr(exploration_cost),
        ptr(exploration_action), ptr(exploration_gradient), ptr(exploration_idx), ptr(selected_cost),
        ptr(selected_action), ptr(selected_gradient), ptr(selected_idx), ptr(search_cost), ptr(search_action),
        ptr(search_gradient), ptr(step_direction_scaled), ptr(search_magnitudes), armijo_threshold_c_1,
        curvature_threshold_c_2, int(strong_wolfe), int(approx_wolfe), n_linesearch, opt_dim, batchsize,
        ptr(step_vec), ptr(rho_buffer), ptr(y_buffer), ptr(s_buffer), ptr(x_0), ptr(grad_0), epsilon, history_m,
        int(stable_mode), ptr(action_step_max), action_dim, int(apply_step_scale), int(overlapped),
        current_stream(best_cost),
    ))


def mppi_update_distribution(
    new_mean, new_cov, new_scale_tril, best_traj, weights, costs, gamma_seq, actions, mean, cov,
    beta: float, step_size_mean: float, step_size_cov: float, kappa: float,
):
    """One launch for ``MPPI._update_distribution`` (reference optim/particle/mppi.py:201-313,
    DIAG_A covariance): softmax weights, blended mean, blended diagonal covariance + its square
    root, best sample.  ``costs`` [problems, particles, cost_horizon]; ``actions``
    [problems, particles, action_horizon, action_dim]; ``best_traj`` / ``weights`` may be None."""
    nb, npart, hc = costs.shape
    ha, d = actions.shape[-2], actions.shape[-1]
    check(load().curobo_hip_mppi_update_distribution(
        ptr(new_mean), ptr(new_cov), ptr(new_scale_tril), ptr(best_traj), ptr(weights), ptr(costs), ptr(gamma_seq),
        ptr(actions), ptr(mean), ptr(cov), nb, npart, hc, ha, d, float(beta), float(step_size_mean),
        float(step_size_cov), float(kappa), current_stream(new_mean),
    ))
