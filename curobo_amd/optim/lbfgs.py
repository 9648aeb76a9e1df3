"""L-BFGS with a parallel Wolfe line search over pre-evaluated step magnitudes.

Mirrors the reference's ``LBFGSOpt`` / ``GradientOptCore`` iteration
(``curobo/_src/optim/gradient/lbfgs.py:156-265``, ``optim/components/gradient_opt_core.py:255-480``,
``optim/gradient/line_search_strategy.py:488-734``; call stack in SURVEY.md section 3.3):

    for each iteration:
        x_set = x_explore + alpha_k * clamp(d)          (prepare_search_points, fused HIP kernel)
        cost, grad = rollout(x_set)                      (num_problems * n_linesearch rollouts)
        selected / exploration / best / converged       (launch_line_search)
        d = L-BFGS two-loop(x_explore, g_explore)        (launch_lbfgs_step)

``inner_iters`` iterations are captured once into a hipGraph (``torch.cuda.CUDAGraph`` is
hipGraph on ROCm) and replayed, like the reference's ``GraphExecutor`` around ``_opt_iters``
(``util/cuda_graph_util.py:144-180``).
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple

import torch

from ..backends import optimization as optimization_hip


@dataclass
class LBFGSOptCfg:
    """Names and defaults follow ``content/configs/task/trajopt/lbfgs_bspline_trajopt.yml:3-40``."""

    num_problems: int = 1
    history: int = 27
    inner_iters: int = 25
    num_iters: int = 100
    line_search_scale: List[float] = field(default_factory=lambda: [0.0, 0.1, 0.5, 1.0])
    line_search_type: str = "approx_wolfe"  # approx_wolfe | wolfe | strong_wolfe
    line_search_c_1: float = 1e-5
    line_search_c_2: float = 0.9
    epsilon: float = 0.01
    stable_mode: bool = True
    step_scale: float = 0.98
    initial_step_scale: float = 0.001
    cost_delta_threshold: float = 0.0
    cost_relative_threshold: float = 0.001
    convergence_iteration: int = 10
    #: reference LBFGSOptCfg.fixed_iters / converged_ratio (optim/gradient/lbfgs.py:61-62, gradient_opt_core.py:305-314):
    #: False = between blocks of ``inner_iters`` iterations the optimiser stops once more than ``converged_ratio`` of the
    #: problems carry the line-search kernel's convergence flag (best cost not improved for ``convergence_iteration``
    #: iterations).  One host read-back per block.  The reference's task configs keep it True.
    fixed_iters: bool = True
    converged_ratio: float = 0.8
    #: line search + two-loop + next candidates in one launch (opt_dim <= 128); False = the three
    #: drop-in launches of the reference's iteration
    fused_tail: bool = True

    def improvement_thresholds(self) -> Tuple[float, float]:
        """(cost_delta_threshold, cost_relative_threshold) as the kernels get them.  Reference optim/gradient/gradient_descent.py:
        68-75 (the base of its LBFGSOptCfg): with a fixed iteration count both are switched off -- they feed the line-search kernel's
        `update_best = cost_delta > delta_threshold && cost_relative > relative_threshold` (line_search_helpers.cuh:33), so a task
        file's `cost_relative_threshold: 1.0` (lbfgs_mpc.yml) would otherwise freeze the best iterate at the seed -- and a relative
        threshold of 1 or more is an error.  (Evaluated at use, not at construction: `fixed_iters` is switched after the fact.)"""
        if self.fixed_iters:
            return 0.0, 0.0
        if self.cost_relative_threshold >= 1.0:
            raise ValueError("cost_relative_threshold must be less than 1.0")
        return float(self.cost_delta_threshold), float(self.cost_relative_threshold)


class LBFGSOpt:
    """``optimize(seed)`` runs ``num_iters`` iterations on ``num_problems`` independent problems.

    ``cost_and_gradient(x[B*NLS, V]) -> (cost[B*NLS], grad[B*NLS, V])`` is the rollout; its
    batch must be ``num_problems * n_linesearch``.  ``action_bounds`` = (low[D], high[D]).
    """

    def __init__(self, cfg: LBFGSOptCfg, cost_and_gradient: Callable, action_horizon: int, action_dim: int,
                 action_bounds: Tuple[torch.Tensor, torch.Tensor], device, use_cuda_graph: bool = True):
        self.cfg = cfg
        self.rollout_fn = cost_and_gradient
        self.action_horizon, self.action_dim = action_horizon, action_dim
        self.opt_dim = action_horizon * action_dim
        self.device = device
        self.use_cuda_graph = use_cuda_graph
        # set by PipelinedLBFGS: this optimiser's launches run next to the rollouts of other seed shards, so its
        # iteration tail takes the form that does not wait for their LDS / registers (one wavefront per problem)
        self.overlapped = False
        # set by a solver whose SEED axis is sharded over the ranks of torch.distributed (IKSolver / TrajOptSolver with
        # global_num_seeds): only then is the convergence exit a collective -- ranks that solve independent problems must not
        # meet in an all-reduce they enter a different number of times
        self.rank_sharded = False
        if self.opt_dim >= 1024:  # reference lbfgs.py:177
            raise ValueError("opt_dim must be < 1024 for the fused L-BFGS step")
        if cfg.history > 31:
            raise ValueError("History_m greater than 31 is not supported")
        self.history = min(cfg.history, self.opt_dim)  # reference lbfgs.py:189-191
        lows, highs = action_bounds
        self._step_max = (cfg.step_scale * torch.abs(highs - lows)).to(device=device, dtype=torch.float32).contiguous()
        self._alphas = torch.tensor(cfg.line_search_scale, device=device, dtype=torch.float32)
        self.n_linesearch = len(cfg.line_search_scale)
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._alloc(cfg.num_problems)

    # reference QuasiNewtonBuffers (optim/components/quasi_newton_buffers.py:23-142) and
    # OptimizationIterationState (optim/components/...): all static, graph-safe.
    def _alloc(self, B: int) -> None:
        d, V, m, N = self.device, self.opt_dim, self.history, self.n_linesearch
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=d, dtype=dt)  # noqa: E731
        self.num_problems = B
        self.y, self.s, self.rho = z(m, B, V), z(m, B, V), z(m, B)
        self.x_0, self.grad_0 = z(B, V), z(B, V)
        self.step_direction = z(B, V)
        self.step_scaled = z(B, V)
        self.x_set = z(B, N, V)
        self.search_cost = z(B, N, 1)
        self.search_gradient = z(B, N, V)
        self.action, self.gradient, self.cost = z(B, V), z(B, V), z(B)
        self.exploration_action, self.exploration_gradient, self.exploration_cost = z(B, V), z(B, V), z(B)
        self.best_action, self.best_cost = z(B, V), z(B)
        self.best_iteration, self.current_iteration = z(B, dt=torch.int16), z(B, dt=torch.int16)
        self.converged = z(B, dt=torch.uint8)
        self.exploration_idx, self.selected_idx = z(B, N, dt=torch.int32), z(B, N, dt=torch.int32)
        self._graph = None

    # ------------------------------------------------------------------ one iteration
    def _evaluate_search_points(self) -> None:
        B, N, V = self.num_problems, self.n_linesearch, self.opt_dim
        cost, grad = self.rollout_fn(self.x_set.view(B * N, V))
        # The rollout returns its own static output buffers; alias them instead of copying
        # (pointers are stable across calls, so this is also what the captured graph sees).
        if cost.data_ptr() != self.search_cost.data_ptr():
            self.search_cost = cost.view(B, N, 1)
        if grad.data_ptr() != self.search_gradient.data_ptr():
            self.search_gradient = grad.view(B, N, V)

    @property
    def _use_fused_tail(self) -> bool:
        return self.cfg.fused_tail and self.opt_dim <= 128

    def _prepare_search_points(self) -> None:
        cfg, B, N, V = self.cfg, self.num_problems, self.n_linesearch, self.opt_dim
        apply_scale = cfg.step_scale != 0.0 and cfg.step_scale != 1.0
        optimization_hip.prepare_search_points(
            self.x_set, self.step_scaled, self.exploration_action, self.step_direction, self._step_max,
            self._alphas, B, N, V, self.action_dim, apply_scale)

    def _opt_step(self) -> None:
        cfg, B, N, V = self.cfg, self.num_problems, self.n_linesearch, self.opt_dim
        apply_scale = cfg.step_scale != 0.0 and cfg.step_scale != 1.0
        if self._use_fused_tail:
            # x_set / step_scaled of this iteration were produced by the previous tail (or by
            # reinitialize): rollout, then one launch for everything on the optimiser side
            self._evaluate_search_points()
            optimization_hip.launch_lbfgs_iteration_tail(
                self.best_cost, self.best_action, self.best_iteration, self.current_iteration, self.converged,
                cfg.convergence_iteration, *cfg.improvement_thresholds(),
                self.exploration_cost, self.exploration_action, self.exploration_gradient,
                self.exploration_idx.view(-1), self.cost, self.action, self.gradient, self.selected_idx.view(-1),
                self.search_cost, self.x_set, self.search_gradient, self.step_scaled, self._alphas,
                cfg.line_search_c_1, cfg.line_search_c_2, cfg.line_search_type == "strong_wolfe",
                cfg.line_search_type == "approx_wolfe", N, V, B, self.step_direction, self.rho, self.y, self.s,
                self.x_0, self.grad_0, cfg.epsilon, self.history, cfg.stable_mode, self._step_max,
                self.action_dim, apply_scale, overlapped=self.overlapped)
            return
        optimization_hip.prepare_search_points(
            self.x_set, self.step_scaled, self.exploration_action, self.step_direction, self._step_max,
            self._alphas, B, N, V, self.action_dim, apply_scale)
        self._evaluate_search_points()
        optimization_hip.launch_line_search(
            self.best_cost, self.best_action, self.best_iteration, self.current_iteration, self.converged,
            cfg.convergence_iteration, *cfg.improvement_thresholds(),
            self.exploration_cost, self.exploration_action, self.exploration_gradient,
            self.exploration_idx.view(-1), self.cost, self.action, self.gradient, self.selected_idx.view(-1),
            self.search_cost, self.x_set, self.search_gradient, self.step_scaled, self._alphas,
            cfg.line_search_c_1, cfg.line_search_c_2, cfg.line_search_type == "strong_wolfe",
            cfg.line_search_type == "approx_wolfe", N, V, B)
        optimization_hip.launch_lbfgs_step(
            self.step_direction, self.rho, self.y, self.s, self.exploration_action, self.exploration_gradient,
            self.x_0, self.grad_0, cfg.epsilon, B, self.history, V, cfg.stable_mode, True)

    def _opt_iters(self) -> None:
        for _ in range(self.cfg.inner_iters):
            self._opt_step()

    # ------------------------------------------------------------------ lifecycle
    def reinitialize(self, seed: torch.Tensor) -> None:
        """reference GradientOptCore._prepare_initial_iteration_state (:402-441): evaluate the seed,
        first direction = -initial_step_scale * gradient, L-BFGS reference point = the seed."""
        B, V = self.num_problems, self.opt_dim
        x = seed.reshape(B, V).to(self.device, torch.float32)
        for t in (self.y, self.s, self.rho):
            t.zero_()
        self.best_cost.fill_(1e10)
        self.best_iteration.zero_()
        self.current_iteration.zero_()
        self.converged.zero_()
        self.exploration_action.copy_(x)
        self.action.copy_(x)
        self.best_action.copy_(x)
        self.x_set.copy_(x.unsqueeze(1).expand(B, self.n_linesearch, V))
        self._evaluate_search_points()
        self.exploration_gradient.copy_(self.search_gradient[:, 0])
        self.exploration_cost.copy_(self.search_cost[:, 0, 0])
        self.gradient.copy_(self.exploration_gradient)
        self.cost.copy_(self.exploration_cost)
        self.best_cost.copy_(self.exploration_cost)
        self.step_direction.copy_(-self.cfg.initial_step_scale * self.exploration_gradient)
        self.x_0.copy_(self.exploration_action)
        self.grad_0.copy_(self.exploration_gradient)
        if self._use_fused_tail:  # candidates of the first iteration (later ones come from the tail)
            self._prepare_search_points()

    def capture(self) -> None:
        """Warm up on a side stream, then record ``inner_iters`` iterations into one hipGraph."""
        saved = [t.clone() for t in self._state_tensors()]
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self._opt_step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._opt_iters()
        for t, s in zip(self._state_tensors(), saved):
            t.copy_(s)
        torch.cuda.synchronize(self.device)

    def make_graph(self, n_iters: int, after=None) -> "torch.cuda.CUDAGraph":
        """a hipGraph of ``n_iters`` iterations (optimiser state is restored after the capture); ``after`` (optional callable)
        is captured behind them"""
        saved = [t.clone() for t in self._state_tensors()]
        self._opt_step()  # warm-up outside the capture
        if after is not None:
            after()
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n_iters):
                self._opt_step()
            if after is not None:
                after()
        for t, s in zip(self._state_tensors(), saved):
            t.copy_(s)
        torch.cuda.synchronize(self.device)
        return g

    def _enough_converged(self) -> bool:
        """reference BestTracker.check_convergence (optim/components/best_tracker.py:109-118); on a seed shard the count is
        taken over all ranks so that every rank stops after the same block"""
        n = torch.count_nonzero(self.converged).to(torch.float32).reshape(1)
        total = float(self.converged.numel())
        import torch.distributed as dist

        if self.rank_sharded and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from ..distributed import all_reduce_sum

            both = all_reduce_sum(torch.cat([n, torch.tensor([total], device=n.device)]))
            n, total = both[:1], float(both[1])
        return float(n.item()) > total * self.cfg.converged_ratio

    def _state_tensors(self):
        return [self.y, self.s, self.rho, self.x_0, self.grad_0, self.step_direction, self.action,
                self.gradient, self.cost, self.exploration_action, self.exploration_gradient,
                self.exploration_cost, self.best_action, self.best_cost, self.best_iteration,
                self.current_iteration, self.converged, self.x_set, self.step_scaled]

    def run_inner(self) -> None:
        """``inner_iters`` iterations (graph replay when enabled)."""
        if self.use_cuda_graph:
            if self._graph is None:
                self.capture()
            self._graph.replay()
        else:
            self._opt_iters()

    def optimize(self, seed: torch.Tensor, num_iters: Optional[int] = None) -> torch.Tensor:
        """``num_iters`` overrides ``cfg.num_iters`` for this call (reference ``update_niters``, used by the trajopt
        solver's finetune passes): whole blocks of ``inner_iters`` replay the graph, a remainder runs eagerly."""
        self.reinitialize(seed)
        if num_iters is None:
            outer, rest = max(1, self.cfg.num_iters // self.cfg.inner_iters), 0
        else:
            outer, rest = divmod(int(num_iters), self.cfg.inner_iters)
        self.iterations_run = 0
        for _ in range(outer):
            self.run_inner()
            self.iterations_run += self.cfg.inner_iters
            if not self.cfg.fixed_iters and self._enough_converged():
                return self.best_action.view(self.num_problems, self.action_horizon, self.action_dim)
        for _ in range(rest):
            self._opt_step()
        self.iterations_run += rest
        return self.best_action.view(self.num_problems, self.action_horizon, self.action_dim)
