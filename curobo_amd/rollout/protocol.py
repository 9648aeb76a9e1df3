"""The Rollout protocol and its result containers -- counterpart of ``curobo/_src/rollout/rollout_protocol.py``
(the members optimisers and solvers rely on: ``evaluate_action``, ``compute_metrics_from_action``, ``update_params``,
``update_batch_size``, action bounds / horizon / dim, ``sum_horizon``) and of the cost containers of
``curobo/_src/rollout/metrics.py`` (``CostCollection`` :23-110, ``CostsAndConstraints.get_sum_cost_and_constraint``
:233-265, ``RolloutResult``, ``RolloutMetrics``).  A rollout is what an optimiser minimises: actions ``[batch,
action_horizon, action_dim]`` in, per-step cost terms out."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Protocol, runtime_checkable

import torch


@dataclass
class CostCollection:
    """named cost terms, each [batch, horizon, k]"""

    values: List[torch.Tensor] = field(default_factory=list)
    names: List[str] = field(default_factory=list)
    weights: Optional[list] = None
    sq_weights: Optional[list] = None

    def add(self, value: torch.Tensor, name: str) -> None:
        self.values.append(value)
        self.names.append(name)

    def is_empty(self) -> bool:
        return len(self.values) == 0

    def sum(self, sum_horizon: bool) -> Optional[torch.Tensor]:
        """sum over the terms (and their last axis) -> [batch, horizon], or [batch] with ``sum_horizon``"""
        if not self.values:
            return None
        total = torch.cat([v.reshape(v.shape[0], v.shape[1], -1) for v in self.values], dim=-1).sum(dim=-1)
        return total.sum(dim=-1) if sum_horizon else total


@dataclass
class CostsAndConstraints:
    costs: CostCollection = field(default_factory=CostCollection)
    constraints: CostCollection = field(default_factory=CostCollection)

    def get_sum_cost(self, sum_horizon: bool = False) -> Optional[torch.Tensor]:
        return self.costs.sum(sum_horizon)

    def get_sum_constraint(self, sum_horizon: bool = False) -> Optional[torch.Tensor]:
        return self.constraints.sum(sum_horizon)

    def get_sum_cost_and_constraint(self, sum_horizon: bool = False) -> torch.Tensor:
        c, k = self.costs.sum(sum_horizon), self.constraints.sum(sum_horizon)
        if c is None:
            return k
        return c if k is None else c + k

    def get_feasible(self, sum_horizon: bool = True) -> Optional[torch.Tensor]:
        k = self.constraints.sum(sum_horizon)
        if k is None:
            c = self.costs.sum(sum_horizon)
            return None if c is None else torch.ones_like(c, dtype=torch.bool)
        return k <= 0.0


@dataclass
class RolloutResult:
    actions: torch.Tensor
    state: object
    costs_and_constraints: CostsAndConstraints


@dataclass
class RolloutMetrics:
    costs_and_constraints: CostsAndConstraints
    feasible: Optional[torch.Tensor] = None
    state: object = None
    convergence: Optional[torch.Tensor] = None
    actions: Optional[torch.Tensor] = None


@runtime_checkable
class Rollout(Protocol):
    """structural type of a rollout (reference rollout_protocol.py:46-174)"""

    sum_horizon: bool

    @property
    def action_dim(self) -> int: ...

    @property
    def action_horizon(self) -> int: ...

    @property
    def action_bound_lows(self) -> torch.Tensor: ...

    @property
    def action_bound_highs(self) -> torch.Tensor: ...

    def evaluate_action(self, act_seq: torch.Tensor, **kwargs) -> RolloutResult: ...

    def compute_metrics_from_action(self, act_seq: torch.Tensor, **kwargs) -> RolloutMetrics: ...

    def update_params(self, **kwargs) -> bool: ...

    def update_batch_size(self, batch_size: int) -> None: ...
