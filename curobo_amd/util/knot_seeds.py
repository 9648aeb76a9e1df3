"""Knot seeds for trajectory optimisation (host side, pure torch).

Mirrors the reference's ``TrajectorySeedGenerator`` (``curobo/_src/util/trajectory_seed_generator.py:
30-120``): constant seeds and straight joint-space lines from the start to one goal configuration
per seed, sampled with ``linspace(0, 1, action_horizon)`` weights (first knot = start, last knot =
goal).  Pinned by the reference's own outputs (``tests/golden/trajectory_seed_golden.npz``).
``TrajOptSolver.seed_knots`` uses interior weights instead (its B-spline boundary knots already pin
the start and the implicit goal state) and perturbs seeds that share a goal.
"""

from __future__ import annotations

import torch


class TrajectorySeedGenerator:
    def __init__(self, action_horizon: int, action_dim: int, device=None, dtype=torch.float32):
        self.action_horizon, self.action_dim = action_horizon, action_dim
        w = torch.linspace(0.0, 1.0, action_horizon, device=device, dtype=dtype)
        # the start weights are the reversed goal weights (not 1 - w: that differs by an ulp), as the reference builds them
        self._interpolation_weights = torch.stack([w.flip(0), w], dim=-1).view(1, action_horizon, 2, 1)

    def generate_constant_seeds(self, constant_position: torch.Tensor, num_seeds: int) -> torch.Tensor:
        """constant_position [B, D] -> [B, num_seeds, action_horizon, D], every knot equal to it"""
        if constant_position.ndim != 2 or constant_position.shape[-1] != self.action_dim:
            raise ValueError(f"constant_position must be [batch, {self.action_dim}], got {tuple(constant_position.shape)}")
        B, D = constant_position.shape
        return constant_position.view(B, 1, 1, D).expand(B, num_seeds, self.action_horizon, D).contiguous()

    def generate_interpolated_seeds(self, start_position: torch.Tensor, goal_position: torch.Tensor, num_seeds: int) -> torch.Tensor:
        """start_position [B, D], goal_position [B, num_seeds, D] -> [B, num_seeds, action_horizon, D]"""
        B, D = start_position.shape
        if D != self.action_dim or goal_position.shape != (B, num_seeds, D):
            raise ValueError(f"expected start [batch, {self.action_dim}] and goal [batch, {num_seeds}, {self.action_dim}], got "
                             f"{tuple(start_position.shape)} and {tuple(goal_position.shape)}")
        w_start, w_goal = self._interpolation_weights[0, :, 0, :], self._interpolation_weights[0, :, 1, :]  # [H, 1] each
        # start-weighted term first, then the goal-weighted one: the same two roundings as the reference's expression
        return w_start * start_position.view(B, 1, 1, D) + w_goal * goal_position.view(B, num_seeds, 1, D)
