"""Device-resident robot tensors, field-for-field the reference's ``KinematicsParams``
(``curobo/_src/robot/types/kinematics_params.py:22-163``) and ``SelfCollisionKinematicsCfg``
(``curobo/_src/robot/types/self_collision_params.py:16-59``)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch

from .loader import RobotModel


@dataclass
class SelfCollisionKinematicsCfg:
    num_spheres: int
    sphere_padding: torch.Tensor  # [S] f32
    collision_pairs: torch.Tensor  # [P,2] i16

    # The reference derives CUDA launch geometry from the pair count (:31-59).  The HIP backend
    # picks its own tiling, the values are kept for signature parity of self_collision_distance.
    @property
    def num_blocks_per_batch(self) -> int:
        return 1

    @property
    def max_threads_per_block(self) -> int:
        return 256


@dataclass
class KinematicsParams:
    fixed_transforms: torch.Tensor
    link_map: torch.Tensor
    joint_map: torch.Tensor
    joint_map_type: torch.Tensor
    joint_offset_map: torch.Tensor
    tool_frame_map: torch.Tensor
    link_chain_data: torch.Tensor
    link_chain_offsets: torch.Tensor
    joint_links_data: torch.Tensor
    joint_links_offsets: torch.Tensor
    joint_affects_endeffector: torch.Tensor
    link_spheres: torch.Tensor
    link_sphere_idx_map: torch.Tensor
    link_masses_com: torch.Tensor
    link_inertias: torch.Tensor
    num_dof: int
    joint_names: List[str]
    tool_frames: List[str]
    joint_limits_position: torch.Tensor
    joint_limits_velocity: torch.Tensor
    self_collision: Optional[SelfCollisionKinematicsCfg] = None
    #: links sorted by tree depth + CSR offsets per level (reference :147-152, _compute_link_levels :255-290)
    link_level_data: Optional[torch.Tensor] = None
    link_level_offsets: Optional[torch.Tensor] = None
    max_level_width: int = 1
    joint_limits_effort: Optional[torch.Tensor] = None  # [D] max |torque| per joint (URDF effort limits)

    @property
    def n_tree_levels(self) -> int:
        return 0 if self.link_level_offsets is None else int(self.link_level_offsets.shape[0]) - 1

    @property
    def num_links(self) -> int:
        return self.link_map.shape[0]

    @property
    def num_pose_links(self) -> int:
        return self.tool_frame_map.shape[0]

    @property
    def num_spheres(self) -> int:
        return self.link_spheres.shape[1]

    @property
    def num_envs(self) -> int:
        return self.link_spheres.shape[0]

    @property
    def device(self):
        return self.fixed_transforms.device

    def validate_shapes(self) -> None:
        """Cross-field consistency of the CSR tables (reference ``KinematicsParams.validate_shapes``,
        robot/types/kinematics_params.py:212-248; the reference's ``KinematicsFusedFunction.forward`` calls
        it before every launch, cuda_ops/kinematics.py:155): sizes of ``link_chain_offsets``,
        ``joint_links_offsets`` and ``joint_affects_endeffector`` against the link / dof / tool-frame counts."""
        L, D, T = int(self.fixed_transforms.shape[0]), int(self.num_dof), len(self.tool_frames)
        if self.link_chain_offsets is not None and self.link_chain_offsets.shape[0] != L + 1:
            raise ValueError(f"link_chain_offsets.size(0) = {self.link_chain_offsets.shape[0]}, expected num_links + 1 = {L + 1}")
        if self.joint_links_offsets is not None and self.joint_links_offsets.shape[0] != D + 1:
            raise ValueError(f"joint_links_offsets.size(0) = {self.joint_links_offsets.shape[0]}, expected num_dof + 1 = {D + 1}")
        if self.joint_affects_endeffector is not None and self.joint_affects_endeffector.numel() != D * T:
            raise ValueError(f"joint_affects_endeffector.numel() = {self.joint_affects_endeffector.numel()}, expected "
                             f"num_dof * n_tool_frames = {D} * {T} = {D * T}")

    @staticmethod
    def from_model(model: RobotModel, device) -> "KinematicsParams":
        def up(a, dtype=None):
            t = torch.as_tensor(a)
            if dtype is not None:
                t = t.to(dtype)
            return t.to(device).contiguous()

        sc = SelfCollisionKinematicsCfg(
            num_spheres=model.num_spheres,
            sphere_padding=up(model.sphere_padding, torch.float32),
            collision_pairs=up(model.collision_pairs, torch.int16),
        )
        from ..backends.rollout import attach_self_lane_lists

        attach_self_lane_lists(sc.collision_pairs, model.num_spheres)  # (host work, once per robot: see the function)
        depth, levels = [], {}
        for k, par in enumerate([int(x) for x in model.link_map]):
            d = 0 if (par < 0 or par == k) else depth[par] + 1
            depth.append(d)
            levels.setdefault(d, []).append(k)
        level_data = [k for d in sorted(levels) for k in levels[d]]
        level_offsets = [0]
        for d in sorted(levels):
            level_offsets.append(level_offsets[-1] + len(levels[d]))
        return KinematicsParams(
            link_level_data=up(level_data, torch.int16),
            link_level_offsets=up(level_offsets, torch.int16),
            max_level_width=max(len(v) for v in levels.values()),
            fixed_transforms=up(model.fixed_transforms, torch.float32),
            link_map=up(model.link_map, torch.int16),
            joint_map=up(model.joint_map, torch.int16),
            joint_map_type=up(model.joint_map_type, torch.int8),
            joint_offset_map=up(model.joint_offset_map, torch.float32),
            tool_frame_map=up(model.tool_frame_map, torch.int16),
            link_chain_data=up(model.link_chain_data, torch.int16),
            link_chain_offsets=up(model.link_chain_offsets, torch.int16),
            joint_links_data=up(model.joint_links_data, torch.int16),
            joint_links_offsets=up(model.joint_links_offsets, torch.int16),
            joint_affects_endeffector=up(model.joint_affects_endeffector, torch.bool),
            link_spheres=up(model.link_spheres, torch.float32),
            link_sphere_idx_map=up(model.link_sphere_idx_map, torch.int16),
            link_masses_com=up(model.link_masses_com, torch.float32),
            link_inertias=up(model.link_inertias, torch.float32),
            num_dof=model.num_dof,
            joint_names=list(model.joint_names),
            tool_frames=list(model.tool_frames),
            joint_limits_position=up(model.joint_limits_position, torch.float32),
            joint_limits_velocity=up(model.joint_limits_velocity, torch.float32),
            self_collision=sc,
            joint_limits_effort=up(model.joint_limits_effort, torch.float32) if getattr(model, "joint_limits_effort", None) is not None else None,
        )
