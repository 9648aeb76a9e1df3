"""``ToolPoseCost`` (reference ``cost/cost_tool_pose.py:25-200``, ``cost/tool_pose_criteria.py``):
goal-set pose cost of the tool frames; ``forward(current_position[B,H,T,3], current_quat[B,H,T,4] (wxyz),
goal_position[G,T,n_goalset,3], goal_quat[G,T,n_goalset,4], idxs_goal[B])`` returns the reference's tuple
``(cost[B,H,2T], linear_distance[B,H,T], angular_distance[B,H,T], goalset_idx[B,H,T])``."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch

from ..hip_ops.cost import ToolPoseDistance


@dataclass
class ToolPoseCostCfg:  # reference ToolPoseCostCfg / ToolPoseCriteria (one criteria set for all links)
    num_links: int
    weight: List[float] = field(default_factory=lambda: [1.0, 1.0])  # position, orientation
    terminal_pose_axes_weight_factor: List[float] = field(default_factory=lambda: [1.0] * 6)
    non_terminal_pose_axes_weight_factor: List[float] = field(default_factory=lambda: [0.0] * 6)
    terminal_pose_convergence_tolerance: List[float] = field(default_factory=lambda: [0.0, 0.0])
    non_terminal_pose_convergence_tolerance: List[float] = field(default_factory=lambda: [0.0, 0.0])
    project_distance_to_goal: bool = False
    use_lie_group: bool = False
    use_grad_input: bool = True

    @property
    def rotation_method(self) -> int:  # reference cost_tool_pose_cfg.py:93-95
        return 1 if self.use_lie_group else 0


class ToolPoseCost:
    def __init__(self, config: ToolPoseCostCfg, device):
        self.config, self.device = config, torch.device(device)
        c, d, T = config, self.device, config.num_links
        t = lambda v: torch.tensor(v, dtype=torch.float32, device=d)  # noqa: E731
        self._weight = t(c.weight)
        self._axes_t = t(c.terminal_pose_axes_weight_factor).repeat(T)
        self._axes_n = t(c.non_terminal_pose_axes_weight_factor).repeat(T)
        self._tol_t = t(c.terminal_pose_convergence_tolerance).repeat(T)
        self._tol_n = t(c.non_terminal_pose_convergence_tolerance).repeat(T)
        self._project = torch.full((T,), int(c.project_distance_to_goal), dtype=torch.uint8, device=d)
        self._shape = None

    def setup_batch_tensors(self, batch_size: int, horizon: int) -> None:
        if self._shape == (batch_size, horizon):
            return
        T, d = self.config.num_links, self.device
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=d, dtype=dt)  # noqa: E731
        self._out_distance = z(batch_size, horizon, 2 * T)
        self._out_position_distance, self._out_rotation_distance = z(batch_size, horizon, T), z(batch_size, horizon, T)
        self._out_goalset_idx = z(batch_size, horizon, T, dt=torch.int32)
        self._out_position_gradient, self._out_rotation_gradient = z(batch_size, horizon, T, 3), z(batch_size, horizon, T, 4)
        self._idx0 = z(batch_size, dt=torch.int32)
        self._shape = (batch_size, horizon)

    def forward(self, current_position: torch.Tensor, current_quat: torch.Tensor, goal_position: torch.Tensor,
                goal_quat: torch.Tensor, idxs_goal: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, ...]:
        T = self.config.num_links
        if self._shape is None or current_position.shape != (*self._shape, T, 3):
            raise ValueError(f"current_position must be (batch, horizon, {T}, 3) matching setup_batch_tensors(), got "
                             f"{tuple(current_position.shape)}")
        if goal_position.shape[1] != T or goal_position.shape[:3] != goal_quat.shape[:3]:
            raise ValueError("goal_position / goal_quat must be [n_goals, num_links, num_goalset, 3 | 4]")
        idx = self._idx0 if idxs_goal is None else idxs_goal.to(torch.int32).contiguous()
        return ToolPoseDistance.apply(
            current_position, current_quat, goal_position.contiguous(), goal_quat.contiguous(), idx, self._weight, self._axes_t,
            self._axes_n, self._tol_t, self._tol_n, self._project, self._out_distance, self._out_position_distance,
            self._out_rotation_distance, self._out_position_gradient, self._out_rotation_gradient, self._out_goalset_idx,
            self.config.use_grad_input, int(goal_position.shape[2]), self.config.rotation_method)

    __call__ = forward
