"""Public collision-checking API -- counterpart of ``curobo.collision_checking``
(``RobotCollisionChecker`` = reference ``curobo/_src/collision/collision_robot_scene.py:26-541``):
joint configurations -> (scene distance per sphere, self-collision distance per point)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from .hip_ops.collision import CollisionBuffer, SphereObstacleCollision, SweptSphereObstacleCollision
from .hip_ops.geometry import SelfCollisionDistance
from .kinematics import Kinematics, KinematicsCfg
from .scene.data import SceneData


@dataclass
class RobotCollisionCheckerCfg:
    """reference RobotCollisionCheckerCfg (collision/collision_robot_scene.py): robot + world + activation distance"""

    kinematics: KinematicsCfg
    scene: Optional[SceneData] = None
    collision_activation_distance: float = 0.0
    scene_weight: float = 1.0
    self_weight: float = 1.0

    @staticmethod
    def load_from_config(robot_config, scene_model=None, collision_activation_distance: float = 0.0, device="cuda:0",
                         assets_root: str = "", **unused) -> "RobotCollisionCheckerCfg":
        """``robot_config``: packaged name (``"franka.yml"``), yaml path or dictionary; ``scene_model``: the reference's
        scene format (``curobo_amd.scene.config``)."""
        import os

        from .scene.config import scene_from_config

        if isinstance(robot_config, dict):
            kin = KinematicsCfg.from_data_dict(robot_config, assets_root=assets_root, device=device)
        elif os.path.exists(str(robot_config)):
            kin = KinematicsCfg.from_robot_yaml_file(robot_config, assets_root or os.path.dirname(os.path.abspath(robot_config)), device=device)
        else:
            kin = KinematicsCfg.from_packaged(str(robot_config).replace(".yml", "").replace(".yaml", ""), device=device)
        scene = scene_from_config(scene_model, device)
        return RobotCollisionCheckerCfg(kin, scene, collision_activation_distance)


class RobotCollisionChecker:
    def __init__(self, kinematics_cfg, scene: Optional[SceneData] = None, activation_distance: float = 0.0,
                 scene_weight: float = 1.0, self_weight: float = 1.0):
        """``RobotCollisionChecker(config: RobotCollisionCheckerCfg)`` (the reference's constructor) or the explicit
        ``(kinematics_cfg, scene, activation_distance, ...)`` form."""
        if isinstance(kinematics_cfg, RobotCollisionCheckerCfg):
            c = kinematics_cfg
            kinematics_cfg, scene, activation_distance = c.kinematics, c.scene, c.collision_activation_distance
            scene_weight, self_weight = c.scene_weight, c.self_weight
        self.kinematics = Kinematics(kinematics_cfg, compute_spheres=True)
        self.scene = scene
        d = kinematics_cfg.kinematics_config.device
        self._w_scene = torch.tensor([scene_weight], device=d)
        self._w_self = torch.tensor([self_weight], device=d)
        self._eta = torch.tensor([activation_distance], device=d)
        self._max_d = torch.tensor([10000.0], device=d)
        self._shape = None

    def update_world(self, scene: SceneData) -> None:
        """reference :93 -- swap the obstacle store (tensors are replicated per GPU)"""
        self.scene = scene

    def _setup(self, b, h):
        if self._shape == (b, h):
            return
        k = self.kinematics.kinematics_config
        d, S = k.device, k.num_spheres
        self._buf = CollisionBuffer.create(b, h, S, d)
        self._self_d = torch.zeros(b, h, 1, device=d)
        self._self_g = torch.zeros(b, h, S, 4, device=d)
        self._sparse = torch.zeros(b, h, S, dtype=torch.uint8, device=d)
        self._pd = torch.zeros(1, device=d)
        self._bbmv = torch.zeros(1, device=d)
        self._bbmi = torch.zeros(2, dtype=torch.int16, device=d)
        self._env = torch.zeros(b, dtype=torch.int32, device=d)
        self._speed_dt = torch.tensor([0.02], device=d)
        self._shape = (b, h)

    def get_scene_self_collision_distance_from_joints(
            self, q: torch.Tensor, env_query_idx: Optional[torch.Tensor] = None,
            sweep: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """q[B,H,D] -> (d_world[B,H,S], d_self[B,H,1]); reference :247-264.  Differentiable in q."""
        if q.ndim == 2:
            q = q.unsqueeze(1)
        b, h, _ = q.shape
        self._setup(b, h)
        state = self.kinematics.compute_kinematics(q)
        sph = state.robot_spheres
        env = self._env if env_query_idx is None else env_query_idx
        sc = self.kinematics.kinematics_config.self_collision
        d_self = SelfCollisionDistance.apply(
            sph, self._self_d, self._self_g, self._pd, self._sparse, self._w_self, sc.sphere_padding,
            sc.collision_pairs, self._bbmv, self._bbmi, sc.num_blocks_per_batch, sc.max_threads_per_block,
            False, True)
        if self.scene is None:
            d_world = torch.zeros(b, h, sph.shape[2], device=q.device)
        elif sweep:
            d_world = SweptSphereObstacleCollision.apply(
                sph, self._buf, self.scene, self._w_scene, self._eta, self._max_d, self._speed_dt, False, env,
                env_query_idx is not None, True)
        else:
            d_world = SphereObstacleCollision.apply(sph, self._buf, self.scene, self._w_scene, self._eta,
                                                    self._max_d, env, env_query_idx is not None, True)
        return d_world, d_self

    def get_scene_self_collision_distance_from_joint_trajectory(
            self, q: torch.Tensor, env_query_idx: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """reference :266-284 (the same evaluation: every trajectory point is checked on its own)"""
        return self.get_scene_self_collision_distance_from_joints(q, env_query_idx)

    def get_bound(self, q: torch.Tensor) -> torch.Tensor:
        """joint-limit violation cost [B,H,D] (reference :286-312, the c-space POSITION cost with unit
        weight, no activation margin)"""
        from .backends import cost as cost_hip

        if q.ndim != 3:
            raise ValueError(f"q must have shape [batch, horizon, dof], got {tuple(q.shape)}")
        b, h, dof = q.shape
        k = self.kinematics.kinematics_config
        d = q.device
        z1, zd = torch.zeros(1, device=d), torch.zeros(dof, device=d)
        big = torch.stack([torch.full((dof,), -1e9, device=d), torch.full((dof,), 1e9, device=d)])
        cost = torch.zeros(b, h, dof, device=d)
        idx0 = torch.zeros(b, dtype=torch.int32, device=d)
        cost_hip.cspace_position_cost(
            cost, None, None, q.detach().contiguous(), None, zd, idx0, k.joint_limits_position.contiguous(), big,
            torch.tensor([1.0, 0.0], device=d), torch.zeros(2, device=d), z1, torch.ones(dof, device=d),
            torch.zeros(2, device=d), zd, zd, idx0, big, z1, False, b, h, dof)
        return cost

    def validate(self, q: torch.Tensor, env_query_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B,H] True where the configuration is inside the joint limits and free of scene and self
        collision (reference :341-372: the three violation sums are exactly zero)."""
        if q.ndim == 2:
            q = q.unsqueeze(1)
        with torch.no_grad():
            d_world, d_self = self.get_scene_self_collision_distance_from_joints(q, env_query_idx)
            d_bound = self.get_bound(q)
        return (d_world.sum(-1) + d_self[..., 0] + d_bound.sum(-1)) == 0.0

    def validate_trajectory(self, q: torch.Tensor, env_query_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.validate(q, env_query_idx)  # reference :404-417

    # ------------------------------------------------------------------ sampling (reference :314-402)
    rejection_ratio = 10

    def _sampler(self):
        if getattr(self, "_halton", None) is None:
            from .solver.seed_ik import HaltonSeeds

            lim = self.kinematics.kinematics_config.joint_limits_position
            self._halton = HaltonSeeds(lim.shape[1], lim[0].contiguous(), lim[1].contiguous(), seed=1312)
        return self._halton

    def sample(self, n: int, mask_valid: bool = True, env_query_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """n joint configurations [n, dof] (Halton points in the joint limits), collision-free when
        ``mask_valid`` -- fewer than n come back when the rejection sampling runs short, as in the reference."""
        q = self._sampler().get_samples(n * self.rejection_ratio if mask_valid else n, bounded=True)
        if mask_valid:
            q = q[self.validate(q.unsqueeze(1), env_query_idx).view(-1)][:n]
        return q

    def sample_trajectory(self, batch: int, horizon: int, mask_valid: bool = True,
                          env_query_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        sh = horizon * self.rejection_ratio if mask_valid else horizon
        q = self._sampler().get_samples(batch * sh, bounded=True).reshape(batch, sh, -1)
        if mask_valid:
            ok = self.validate_trajectory(q, env_query_idx)
            q = torch.cat([q[i][ok[i]][:horizon].unsqueeze(0) for i in range(batch)])
        return q
