"""The reference's optimiser call surface on top of the HIP iteration loop (``curobo_amd.optim.lbfgs``).

``curobo.optim.LBFGSOpt`` is constructed as ``LBFGSOpt(config, rollout_list, use_cuda_graph)`` from a list of
Rollout-protocol objects (reference ``curobo/_src/optim/gradient/lbfgs.py:156-265``; the base class evaluates
``rollout.evaluate_action(x)`` and back-propagates ``cost.backward(gradient=ones)`` for the gradient,
``optim/components/gradient_opt_core.py:445-480``).  This adapter keeps that contract:

* a rollout that offers ``cost_and_gradient(x[B, V]) -> (cost[B], grad[B, V])`` (the HIP rollouts: explicit VJP,
  fused launch) is called directly;
* any other Rollout-protocol object (``RosenbrockRollout``, user rollouts written in torch) goes through autograd
  exactly as in the reference.

Line search + two-loop run in the HIP kernels either way; ``use_cuda_graph`` replays ``inner_iters`` iterations from
a hipGraph.  Config field names are the reference's ``LBFGSOptCfg`` (lbfgs.py:36-153, gradient_opt_core.py:49-140).
"""

from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .lbfgs import LBFGSOpt as _HipLBFGS
from .lbfgs import LBFGSOptCfg as _HipLBFGSCfg


@dataclass
class LBFGSOptCfg:
    num_problems: int = 1
    num_iters: int = 100
    inner_iters: int = 25
    history: int = 27
    epsilon: float = 0.01
    stable_mode: bool = True
    step_scale: float = 0.98
    initial_step_scale: float = 0.001
    line_search_scale: List[float] = field(default_factory=lambda: [0.0, 0.1, 0.5, 1.0])
    line_search_type: str = "approx_wolfe"
    line_search_c_1: float = 1e-5
    line_search_c_2: float = 0.9
    cost_delta_threshold: float = 0.0
    cost_relative_threshold: float = 0.001
    cost_convergence: float = 0.0
    convergence_iteration: int = 10
    fixed_iters: bool = True       # reference optim/gradient/lbfgs.py:61-62
    converged_ratio: float = 0.8
    solver_type: str = "lbfgs"
    # the reference's kernel switches: the HIP backend has no torch fallback, the flags are accepted and ignored
    use_cuda_kernel_step_direction: bool = True
    use_cuda_kernel_shared_buffers: bool = True
    use_cuda_kernel_line_search: bool = True
    sync_cuda_time: bool = True
    store_debug: bool = False

    def to_hip(self) -> _HipLBFGSCfg:
        return _HipLBFGSCfg(
            num_problems=self.num_problems, history=self.history, inner_iters=max(1, min(self.inner_iters, self.num_iters)),
            num_iters=self.num_iters, line_search_scale=list(self.line_search_scale), line_search_type=self.line_search_type,
            line_search_c_1=self.line_search_c_1, line_search_c_2=self.line_search_c_2, epsilon=self.epsilon,
            stable_mode=self.stable_mode, step_scale=self.step_scale, initial_step_scale=self.initial_step_scale,
            cost_delta_threshold=self.cost_delta_threshold, cost_relative_threshold=self.cost_relative_threshold,
            convergence_iteration=self.convergence_iteration, fixed_iters=self.fixed_iters, converged_ratio=self.converged_ratio)


class LBFGSOpt:
    """``LBFGSOpt(config, rollout_list, use_cuda_graph=False)``; ``reinitialize(action)`` then ``optimize(action)``
    -> best action ``[num_problems, action_horizon, action_dim]`` (reference lbfgs.py:371-421)."""

    def __init__(self, config: LBFGSOptCfg, rollout_list: list, use_cuda_graph: bool = False):
        if not rollout_list:
            raise ValueError("LBFGSOpt needs at least one rollout")
        self.config, self.rollout_list, self.use_cuda_graph = config, list(rollout_list), use_cuda_graph
        self.rollout_fn = self.rollout_list[0]
        self.solve_time = 0.0
        self._opt: Optional[_HipLBFGS] = None
        self._build(config.num_problems)

    # ---- shapes
    @property
    def action_horizon(self) -> int:
        return int(self.rollout_fn.action_horizon)

    @property
    def action_dim(self) -> int:
        return int(self.rollout_fn.action_dim)

    @property
    def opt_dim(self) -> int:
        return self.action_horizon * self.action_dim

    @property
    def num_problems(self) -> int:
        return self.config.num_problems

    def _cost_and_gradient(self, x: torch.Tensor):
        ro = self.rollout_fn
        if hasattr(ro, "cost_and_gradient"):
            return ro.cost_and_gradient(x)
        # Rollout-protocol object: autograd, as the reference's _compute_cost_constraint_and_gradient
        B = x.shape[0]
        with torch.enable_grad():
            xg = x.detach().view(B, self.action_horizon, self.action_dim).clone().requires_grad_(True)
            res = ro.evaluate_action(xg)
            cost = res.costs_and_constraints.get_sum_cost_and_constraint(sum_horizon=True)
            if cost.ndim > 1:
                cost = cost.reshape(B, -1).sum(dim=-1)
            (grad,) = torch.autograd.grad(cost, xg, grad_outputs=torch.ones_like(cost))
        self._cost_buf.copy_(cost.detach())
        self._grad_buf.copy_(grad.reshape(B, -1))
        return self._cost_buf, self._grad_buf

    def _build(self, num_problems: int) -> None:
        self.config.num_problems = num_problems
        ro = self.rollout_fn
        lows, highs = ro.action_bound_lows, ro.action_bound_highs
        device = lows.device
        nls = len(self.config.line_search_scale)
        if hasattr(ro, "update_batch_size"):
            ro.update_batch_size(num_problems * nls)
        self._cost_buf = torch.zeros(num_problems * nls, device=device)
        self._grad_buf = torch.zeros(num_problems * nls, self.opt_dim, device=device)
        self._opt = _HipLBFGS(self.config.to_hip(), self._cost_and_gradient, self.action_horizon, self.action_dim,
                              (lows.reshape(-1)[: self.action_dim], highs.reshape(-1)[: self.action_dim]), device,
                              use_cuda_graph=self.use_cuda_graph)

    # ---- reference call surface
    def update_num_problems(self, num_problems: int) -> None:
        if num_problems != self.config.num_problems or self._opt is None:
            self._build(num_problems)

    def update_rollout_params(self, **kwargs) -> bool:
        return all(bool(r.update_params(**kwargs)) for r in self.rollout_list)

    def reset_cuda_graph(self) -> None:
        self._opt._graph = None

    def reset_seed(self) -> None:
        for r in self.rollout_list:
            if hasattr(r, "reset_seed"):
                r.reset_seed()

    def reinitialize(self, action: torch.Tensor, **kwargs) -> None:
        if action.shape[0] != self.config.num_problems:
            self.update_num_problems(int(action.shape[0]))
        self._opt.reinitialize(action.reshape(self.config.num_problems, self.action_horizon, self.action_dim))

    def optimize(self, seed_action: torch.Tensor, **kwargs) -> torch.Tensor:
        t0 = time.perf_counter()
        if seed_action.shape[0] != self.config.num_problems:
            self.update_num_problems(int(seed_action.shape[0]))
        out = self._opt.optimize(seed_action.reshape(self.config.num_problems, self.action_horizon, self.action_dim))
        if self.config.sync_cuda_time and out.is_cuda:
            torch.cuda.synchronize(out.device)
        self.solve_time = time.perf_counter() - t0
        return out

    # results of the last solve (reference OptimizationIterationState members)
    @property
    def best_cost(self) -> torch.Tensor:
        return self._opt.best_cost

    @property
    def best_action(self) -> torch.Tensor:
        return self._opt.best_action.view(self.config.num_problems, self.action_horizon, self.action_dim)

    @property
    def best_iteration(self) -> torch.Tensor:
        return self._opt.best_iteration
