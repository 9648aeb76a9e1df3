"""``SelfCollisionCost`` / ``SceneCollisionCost`` (reference ``cost/cost_self_collision.py:20-150``,
``cost/cost_scene_collision.py:22-240``)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from ..hip_ops.collision import CollisionBuffer, SphereObstacleCollision, SweptSphereObstacleCollision
from ..hip_ops.geometry import SelfCollisionDistance
from ..robot.kinematics_params import SelfCollisionKinematicsCfg
from ..scene.data import SceneData


@dataclass
class SelfCollisionCostCfg:  # reference SelfCollisionCostCfg
    self_collision_kin_config: SelfCollisionKinematicsCfg
    weight: float = 5000.0
    store_pair_distance: bool = False
    use_grad_input: bool = True
    convert_to_binary: bool = False


class SelfCollisionCost:
    """``forward(robot_spheres[B,H,S,4]) -> cost[B,H]`` = weight * largest pair penetration (0 when free)."""

    def __init__(self, config: SelfCollisionCostCfg, device):
        self.config, self.device = config, torch.device(device)
        self._weight = torch.tensor([config.weight], device=self.device)
        self._shape = None

    def setup_batch_tensors(self, batch_size: int, horizon: int) -> None:
        if self._shape == (batch_size, horizon):
            return
        k, d = self.config.self_collision_kin_config, self.device
        S, P = k.num_spheres, k.collision_pairs.shape[0]
        self._out_distance = torch.zeros(batch_size, horizon, 1, device=d)
        self._out_grad = torch.zeros(batch_size, horizon, S, 4, device=d)
        self._sparse_sphere_idx = torch.zeros(batch_size, horizon, S, dtype=torch.uint8, device=d)
        self._pair_distance = torch.zeros(batch_size, horizon, P if self.config.store_pair_distance else 1, device=d)
        self._block_batch_max_value = torch.zeros(1, device=d)
        self._block_batch_max_index = torch.zeros(2, dtype=torch.int16, device=d)
        self._shape = (batch_size, horizon)

    def validate_input(self, robot_spheres: torch.Tensor) -> None:
        k = self.config.self_collision_kin_config
        if self._shape is None or robot_spheres.shape != (*self._shape, k.num_spheres, 4):
            raise ValueError(f"robot_spheres.shape must be (batch_size, horizon, num_spheres, 4) = {self._shape} x "
                             f"{k.num_spheres} x 4 (call setup_batch_tensors), got {tuple(robot_spheres.shape)}")

    def forward(self, robot_spheres: torch.Tensor) -> torch.Tensor:
        self.validate_input(robot_spheres)
        k, c = self.config.self_collision_kin_config, self.config
        dist = SelfCollisionDistance.apply(
            robot_spheres, self._out_distance, self._out_grad, self._pair_distance, self._sparse_sphere_idx, self._weight,
            k.sphere_padding, k.collision_pairs, self._block_batch_max_value, self._block_batch_max_index,
            k.num_blocks_per_batch, k.max_threads_per_block, c.store_pair_distance, c.use_grad_input)
        dist = dist.view(*self._shape)
        if c.convert_to_binary:  # reference :126-128
            dist = torch.clamp(dist, max=1.0)
            dist = torch.where(dist > 0, dist + 1.0, dist)
        return dist

    __call__ = forward

    def update_weight(self, weight: float) -> None:
        self._weight.fill_(weight)


@dataclass
class SceneCollisionCostCfg:  # reference SceneCollisionCostCfg
    scene: SceneData
    num_spheres: int
    weight: float = 5000.0
    activation_distance: float = 0.01
    use_sweep: bool = False
    use_speed_metric: bool = False
    sum_distance: bool = True
    convert_to_binary: bool = False
    use_grad_input: bool = True


class SceneCollisionCost:
    """``forward(robot_spheres[B,H,S,4], idxs_env_query, trajectory_dt) -> cost[B,H]`` (sum, or max when
    ``sum_distance`` is off, over the spheres of a point; ``convert_to_binary`` as the reference's
    ``jit_weight_collision``)."""

    def __init__(self, config: SceneCollisionCostCfg, device):
        self.config, self.device = config, torch.device(device)
        d = self.device
        self._weight = torch.tensor([config.weight], device=d)
        self._eta = torch.tensor([config.activation_distance], device=d)
        self._max_distance = torch.tensor([10000.0], device=d)
        self._shape = None

    def setup_batch_tensors(self, batch_size: int, horizon: int) -> None:
        if self._shape == (batch_size, horizon):
            return
        self._collision_buffer = CollisionBuffer.create(batch_size, horizon, self.config.num_spheres, self.device)
        self._env0 = torch.zeros(batch_size, dtype=torch.int32, device=self.device)
        self._dt = torch.full((1,), 0.02, device=self.device)
        self._shape = (batch_size, horizon)

    def validate_input(self, robot_spheres, idxs_env_query=None, trajectory_dt=None) -> None:
        if self._shape is None or robot_spheres.shape != (*self._shape, self.config.num_spheres, 4):
            raise ValueError("robot_spheres.shape must be equal to (batch_size, horizon, num_spheres, 4); call "
                             f"setup_batch_tensors().  Got {tuple(robot_spheres.shape)}")
        if idxs_env_query is not None and idxs_env_query.shape != (self._shape[0],):
            raise ValueError("env_query_idx.shape must be equal to (batch_size,)")
        if self.config.use_sweep and trajectory_dt is None:
            raise ValueError("trajectory_dt must be set if use_sweep is True")

    def forward(self, robot_spheres: torch.Tensor, idxs_env_query: Optional[torch.Tensor] = None,
                trajectory_dt: Optional[torch.Tensor] = None) -> torch.Tensor:
        robot_spheres = getattr(robot_spheres, "robot_spheres", robot_spheres)  # a KinematicsState is accepted too
        self.validate_input(robot_spheres, idxs_env_query, trajectory_dt)
        c = self.config
        env = self._env0 if idxs_env_query is None else idxs_env_query
        multi = idxs_env_query is not None
        if c.use_sweep:
            if c.convert_to_binary:
                raise NotImplementedError("convert_to_binary is not implemented for the swept cost (as in the reference)")
            dt = trajectory_dt.reshape(-1)[:1].to(self.device, torch.float32) if trajectory_dt is not None else self._dt
            d = SweptSphereObstacleCollision.apply(robot_spheres, self._collision_buffer, c.scene, self._weight, self._eta,
                                                   self._max_distance, dt, c.use_speed_metric, env, multi, c.use_grad_input)
        else:
            d = SphereObstacleCollision.apply(robot_spheres, self._collision_buffer, c.scene, self._weight, self._eta,
                                              self._max_distance, env, multi, c.use_grad_input)
        out = d.sum(-1) if c.sum_distance else d.max(-1)[0]
        if c.convert_to_binary:
            out = torch.where(out > 0, out + 1.0, out)
        return out

    __call__ = forward

    def get_gradient_buffer(self) -> torch.Tensor:
        return self._collision_buffer.gradient

    def update_weight(self, weight: float) -> None:
        self._weight.fill_(weight)
