"""Chain of optimisers: every enabled stage is seeded with the best action of the previous one.

Mirrors the reference's ``MultiStageOptimizer`` (``curobo/_src/optim/multi_stage_optimizer.py:24-295``;
the reference's trajopt / IK solvers chain a particle stage (MPPI) into L-BFGS this way).  Stages
only need ``optimize(seed) -> action``, ``action_horizon``, ``action_dim`` and a ``num_problems``
(``cfg.num_problems``); :class:`LBFGSOpt`, :class:`PipelinedLBFGS` and :class:`MPPI` qualify.
"""

from __future__ import annotations

import time
from typing import List

import torch


class MultiStageOptimizer:
    def __init__(self, optimizers: List):
        if not optimizers:
            raise ValueError("MultiStageOptimizer needs at least one optimizer")
        self.optimizers = list(optimizers)
        self._stage_enabled = [True] * len(self.optimizers)
        self.cfg = self.optimizers[-1].cfg  # the last stage defines the exposed configuration
        self.opt_dt = 0.0
        self._enabled = True

    # -- reference properties (:60-94)
    @property
    def enabled(self) -> bool:
        return self._enabled

    def enable(self) -> None:
        self._enabled = True

    def disable(self) -> None:
        self._enabled = False

    def enable_stage(self, index: int, enabled: bool = True) -> None:
        self._stage_enabled[index] = enabled

    @property
    def action_horizon(self) -> int:
        return self.optimizers[-1].action_horizon

    @property
    def action_dim(self) -> int:
        return self.optimizers[-1].action_dim

    @property
    def opt_dim(self) -> int:
        return self.action_horizon * self.action_dim

    @property
    def solver_names(self) -> List[str]:
        return [type(o).__name__ for o in self.optimizers]

    @property
    def solve_time(self) -> float:
        return self.opt_dt

    # -- reference optimize / _opt_iters (:96-188)
    def optimize(self, seed_action: torch.Tensor) -> torch.Tensor:
        t0 = time.perf_counter()
        action = seed_action
        for opt, on in zip(self.optimizers, self._stage_enabled):
            if not on or not getattr(opt, "enabled", True):
                continue
            n = opt.cfg.num_problems
            action = opt.optimize(action.reshape(n, opt.action_horizon, opt.action_dim))
        out = action.reshape(self.cfg.num_problems, self.action_horizon, self.action_dim)
        self.opt_dt = time.perf_counter() - t0  # host time of the launches (the caller synchronises)
        return out

    def reinitialize(self, seed_action: torch.Tensor) -> None:
        for opt in self.optimizers:
            if hasattr(opt, "reinitialize"):
                opt.reinitialize(seed_action)

    def reset_cuda_graph(self) -> None:
        for opt in self.optimizers:
            if hasattr(opt, "_graph"):
                opt._graph = None
