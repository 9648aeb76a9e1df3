"""``RosenbrockRollout``: the canonical non-convex test function as a rollout, for exercising optimisers without a
robot (counterpart of ``curobo.rollout.RosenbrockRollout``, reference ``curobo/_src/rollout/rollout_rosenbrock.py``:
f(x) = sum_i (a - x_i)^2 + b (x_{i+1} - x_i^2)^2 over consecutive coordinates, action bounds [-1.5, 2.0])."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from ..types import DeviceCfg, JointState
from .protocol import CostsAndConstraints, RolloutMetrics, RolloutResult


@dataclass
class RosenbrockCfg:
    device_cfg: DeviceCfg = field(default_factory=DeviceCfg)
    a: float = 1.0
    b: float = 100.0
    dimensions: int = 2
    time_horizon: int = 1
    time_action_horizon: int = 1
    sum_horizon: bool = False
    sampler_seed: int = 1312

    @classmethod
    def create(cls, config_dict: Dict, device_cfg: Optional[DeviceCfg] = None) -> "RosenbrockCfg":
        keys = ("a", "b", "dimensions", "time_horizon", "time_action_horizon", "sum_horizon", "sampler_seed")
        return cls(device_cfg=device_cfg or DeviceCfg(), **{k: config_dict[k] for k in keys if k in config_dict})


class RosenbrockRollout:
    def __init__(self, config: Optional[RosenbrockCfg] = None, use_cuda_graph: bool = False):
        config = config or RosenbrockCfg()
        self.config, self.device_cfg = config, config.device_cfg
        self.a, self.b, self.dimensions = config.a, config.b, config.dimensions
        self.sum_horizon = config.sum_horizon
        kw = self.device_cfg.as_torch_dict()
        self._lows = torch.full((config.dimensions,), -1.5, **kw)
        self._highs = torch.full((config.dimensions,), 2.0, **kw)
        self._batch_size = 1
        self._gen = torch.Generator(device="cpu").manual_seed(config.sampler_seed)

    # ---- protocol members
    @property
    def action_dim(self) -> int:
        return self.dimensions

    @property
    def action_horizon(self) -> int:
        return self.config.time_action_horizon

    @property
    def horizon(self) -> int:
        return self.config.time_horizon

    @property
    def action_bound_lows(self) -> torch.Tensor:
        return self._lows

    @property
    def action_bound_highs(self) -> torch.Tensor:
        return self._highs

    @property
    def action_bounds(self) -> torch.Tensor:
        return torch.stack([self._lows, self._highs])

    @property
    def batch_size(self) -> int:
        return self._batch_size

    @property
    def dt(self) -> float:
        return 1.0

    def update_batch_size(self, batch_size: int) -> None:
        self._batch_size = batch_size

    def update_params(self, a: Optional[float] = None, b: Optional[float] = None, **kwargs) -> bool:
        self.a = self.a if a is None else a
        self.b = self.b if b is None else b
        return True

    def update_dt(self, dt, **kwargs) -> bool:
        return True

    def reset(self, **kwargs) -> bool:
        return True

    def reset_shape(self) -> bool:
        return True

    def reset_cuda_graph(self) -> bool:
        return True

    def reset_seed(self) -> None:
        self._gen.manual_seed(self.config.sampler_seed)

    def sample_random_actions(self, n: int = 0, bounded: bool = True) -> torch.Tensor:
        u = torch.rand(n, self.dimensions, generator=self._gen).to(**self.device_cfg.as_torch_dict())
        return self._lows + u * (self._highs - self._lows)

    def get_initial_action(self, use_random: bool = True, use_zero: bool = False, **kwargs) -> torch.Tensor:
        n, h = self._batch_size or 1, self.action_horizon
        if use_random:
            return self.sample_random_actions(n * h).view(n, h, self.dimensions)
        return torch.zeros(n, h, self.dimensions, **self.device_cfg.as_torch_dict())

    # ---- cost
    def _cost(self, x: torch.Tensor) -> CostsAndConstraints:
        c = (self.a - x[..., :-1]) ** 2 + self.b * (x[..., 1:] - x[..., :-1] ** 2) ** 2
        cc = CostsAndConstraints()
        cc.costs.add(c.sum(dim=-1, keepdim=True), "rosenbrock")
        return cc

    def evaluate_action(self, act_seq: torch.Tensor, **kwargs) -> RolloutResult:
        self._batch_size = act_seq.shape[0]
        return RolloutResult(actions=act_seq, state=JointState.from_position(act_seq), costs_and_constraints=self._cost(act_seq))

    def compute_metrics_from_action(self, act_seq: torch.Tensor, **kwargs) -> RolloutMetrics:
        cc = self._cost(act_seq)
        return RolloutMetrics(costs_and_constraints=cc, feasible=cc.get_feasible(), state=JointState.from_position(act_seq),
                              convergence=cc.get_sum_cost(sum_horizon=False), actions=act_seq)

    def compute_metrics_from_state(self, state: JointState, **kwargs) -> RolloutMetrics:
        return self.compute_metrics_from_action(state.position)

    def get_all_cost_components(self):
        return {}
