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
    link_names: Optional[List[str]] = None  # name of link k (index into fixed_transforms / link_sphere_idx_map values)
    reference_link_spheres: Optional[torch.Tensor] = None  # the spheres as loaded: what enable / reset restore
    grasp_contact_link_names: Optional[List[str]] = None

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

    # ---- collision spheres of a link on / off, inertial parameters: in place, so captured graphs (which hold the pointers of
    # ``link_spheres`` / ``link_masses_com`` / ``link_inertias``) see the change on their next replay
    def _link_index(self, link_name: str) -> int:
        if self.link_names is None or link_name not in self.link_names:
            raise ValueError(f"{link_name} not found in the links of the robot")
        return self.link_names.index(link_name)

    def get_sphere_index_from_link_name(self, link_name: str) -> torch.Tensor:
        """indices of the spheres attached to ``link_name`` (reference kinematics_params.py:get_sphere_index_from_link_name)"""
        return torch.nonzero(self.link_sphere_idx_map.to(torch.int64) == self._link_index(link_name)).view(-1)

    def disable_link_spheres(self, link_name: str) -> None:
        """radius -100 in every sphere set: the collision kernels skip spheres with a negative radius (reference :558-567)"""
        self.link_spheres[:, self.get_sphere_index_from_link_name(link_name), 3] = -100.0

    def enable_link_spheres(self, link_name: str) -> None:
        """radii back from the loaded spheres; positions stay (reference :569-582)"""
        idx = self.get_sphere_index_from_link_name(link_name)
        self.link_spheres[:, idx, 3] = self.reference_link_spheres[:, idx, 3]

    def reset_link_spheres(self, link_name: str) -> None:
        idx = self.get_sphere_index_from_link_name(link_name)
        self.link_spheres[:, idx, :] = self.reference_link_spheres[:, idx, :]

    def update_link_inertial(self, link_name: str, mass: Optional[float] = None, com=None, inertia=None) -> None:
        """mass [kg], centre of mass [3] in the link frame, inertia [ixx, iyy, izz, ixy, ixz, iyz] of one link (reference
        robot/dynamics/dynamics.py:401-428)"""
        if mass is None and com is None and inertia is None:
            raise ValueError("At least one property (mass, com, or inertia) must be provided")
        k = self._link_index(link_name)
        f = lambda v: torch.as_tensor(v, dtype=torch.float32).reshape(-1).to(self.device)  # noqa: E731
        if mass is not None:
            self.link_masses_com[k, 3] = float(mass)
        if com is not None:
            self.link_masses_com[k, :3] = f(com)[:3]
        if inertia is not None:
            self.link_inertias[k, :6] = f(inertia)[:6]

    def update_links_inertial(self, link_properties) -> None:
        if not link_properties:
            raise ValueError("link_properties dictionary cannot be empty")
        for name, props in link_properties.items():
            if not props:
                raise ValueError(f"No properties specified for link '{name}'")
            self.update_link_inertial(name, mass=props.get("mass"), com=props.get("com"), inertia=props.get("inertia"))

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
            link_names=list(model.link_names),
            reference_link_spheres=up(model.link_spheres, torch.float32).clone(),
            grasp_contact_link_names=None if getattr(model, "grasp_contact_link_names", None) is None else list(model.grasp_contact_link_names),
        )
