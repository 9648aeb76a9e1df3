"""Public kinematics API -- counterpart of ``curobo.kinematics`` (reference
``curobo/_src/robot/kinematics/kinematics.py:38-198``, ``kinematics_cfg.py:68-213``,
``kinematics_state.py:15-35``): differentiable batched FK over the URDF tree with collision
spheres, tool-frame poses, geometric Jacobian and centre of mass."""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Union

import torch

from .hip_ops.kinematics import KinematicsFusedFunction
from .robot import RobotModel, load_packaged_robot, load_robot_model
from .robot.kinematics_params import KinematicsParams
from .types import _FramePoseMembers


@dataclass
class ToolPose(_FramePoseMembers):
    """positions [B,H,T,3] and quaternions [B,H,T,4] (wxyz) of the tool frames (reference ``ToolPose``, types/tool_pose.py:23-180;
    ``get_link_pose`` / ``to_dict`` / ``reorder_links`` / indexing / copies come with ``_FramePoseMembers``)."""

    tool_frames: List[str]
    position: torch.Tensor
    quaternion: torch.Tensor

    @property
    def batch_size(self) -> int:
        return int(self.position.shape[0])

    @property
    def horizon(self) -> int:
        return int(self.position.shape[1])

    @property
    def num_links(self) -> int:
        return int(self.position.shape[2])

    @property
    def shape(self):
        return self.position.shape

    @property
    def device(self):
        return self.position.device

    def as_goal(self, ordered_tool_frames: Optional[List[str]] = None):
        """the poses as goals for the solvers (reference ToolPose.as_goal, _src/types/tool_pose.py:165-179): a goal-set
        axis of one is added -> GoalToolPose [batch, horizon, T, 1, 3 | 4]"""
        from .types import GoalToolPose

        p, q, frames = self.position.detach(), self.quaternion.detach(), list(self.tool_frames)
        if ordered_tool_frames:
            order = [frames.index(f) for f in ordered_tool_frames]
            p, q, frames = p[:, :, order], q[:, :, order], list(ordered_tool_frames)
        # copies: the kinematics front end re-uses its output buffers on the next call
        return GoalToolPose(frames, p.unsqueeze(3).clone(), q.unsqueeze(3).clone())


@dataclass
class JointLimits:
    """reference JointLimits (robot/types/joint_limits.py): position / velocity [2, dof] (lower, upper), effort [dof]"""

    joint_names: List[str]
    position: torch.Tensor
    velocity: torch.Tensor
    effort: Optional[torch.Tensor] = None


@dataclass
class KinematicsState:
    tool_poses: ToolPose
    tool_jacobians: Optional[torch.Tensor]  # [B,H,T,6,D]
    robot_spheres: Optional[torch.Tensor]  # [B,H,S,4]
    robot_com: Optional[torch.Tensor]  # [B,H,4]


@dataclass
class KinematicsCfg:
    kinematics_config: KinematicsParams
    model: RobotModel

    @staticmethod
    def from_robot_yaml_file(robot_yaml: str, assets_root: str, device="cuda:0", num_envs: int = 1) -> "KinematicsCfg":
        model = load_robot_model(robot_yaml, assets_root, num_envs=num_envs)
        return KinematicsCfg(KinematicsParams.from_model(model, torch.device(device)), model)

    @staticmethod
    def from_data_dict(data_dict: dict, assets_root: str = "", tool_frames: Optional[list] = None, device="cuda:0",
                       num_envs: int = 1) -> "KinematicsCfg":
        """reference KinematicsCfg.from_data_dict (robot/kinematics/kinematics_cfg.py:186-211): the
        ``kinematics`` section of a robot configuration as a dictionary (``urdf_path`` relative to
        ``assets_root``, or absolute)."""
        import copy
        import os

        from .robot.loader import build_robot_model
        from .robot.urdf import load_urdf

        cfg = copy.deepcopy(data_dict.get("robot_cfg", data_dict))
        cfg = cfg.get("kinematics", cfg)
        if tool_frames is not None:
            cfg["tool_frames"] = list(tool_frames)
        model = build_robot_model(cfg, load_urdf(os.path.join(assets_root, cfg["urdf_path"])), num_envs=num_envs)
        return KinematicsCfg(KinematicsParams.from_model(model, torch.device(device)), model)

    @staticmethod
    def from_xrdf(xrdf_path: str, urdf_path: str, tool_frames: Optional[list] = None, device="cuda:0", num_envs: int = 1) -> "KinematicsCfg":
        """an XRDF file next to its URDF (reference KinematicsCfg.from_robot_yaml_file with ``urdf_path`` for ``*.xrdf``,
        robot/kinematics/kinematics_cfg.py:120-160 -> util/xrdf_util.py convert_xrdf_to_curobo)"""
        from .robot.xrdf import convert_xrdf_to_config

        return KinematicsCfg.from_data_dict(convert_xrdf_to_config(xrdf_path, urdf_path), tool_frames=tool_frames, device=device, num_envs=num_envs)

    @staticmethod
    def from_basic_urdf(urdf_path: str, base_link: str, tool_frames: list, device="cuda:0") -> "KinematicsCfg":
        """reference KinematicsCfg.from_basic_urdf (:68-88): kinematics only (no collision spheres)."""
        return KinematicsCfg.from_data_dict({"urdf_path": urdf_path, "base_link": base_link, "tool_frames": list(tool_frames)},
                                            device=device)

    @staticmethod
    def from_packaged(name: str, device="cuda:0") -> "KinematicsCfg":
        model = load_packaged_robot(name)
        return KinematicsCfg(KinematicsParams.from_model(model, torch.device(device)), model)


class Kinematics:
    def __init__(self, config: KinematicsCfg, compute_jacobian: bool = False, compute_spheres: bool = True,
                 compute_com: bool = False):
        self.config = config
        self.kinematics_config = config.kinematics_config
        self.compute_jacobian, self.compute_spheres, self.compute_com = compute_jacobian, compute_spheres, compute_com
        self._shape = None
        self._buffers = None
        self._env_idx = None

    @property
    def joint_names(self):
        return self.kinematics_config.joint_names

    @property
    def tool_frames(self):
        return self.kinematics_config.tool_frames

    # ---- reference members (robot/kinematics/kinematics.py:200-420): sizes, defaults, limits, joint-state bookkeeping
    @property
    def dof(self) -> int:
        return self.kinematics_config.num_dof

    def get_dof(self) -> int:
        return self.dof

    @property
    def base_link(self) -> str:
        return self.config.model.base_link

    @property
    def total_spheres(self) -> int:
        return self.kinematics_config.num_spheres

    @property
    def default_joint_position(self) -> torch.Tensor:
        from .workloads import start_configuration

        return torch.as_tensor(start_configuration(self.config.model), dtype=torch.float32, device=self.kinematics_config.device)

    @property
    def default_joint_state(self):
        from .types import JointState

        return JointState.from_position(self.default_joint_position, joint_names=self.joint_names)

    @property
    def lock_jointstate(self):
        """the joints the robot file locks, at their locked values (reference ``lock_jointstate``)"""
        from .types import JointState

        lock = self.config.model.lock_joints or {}
        return JointState.from_position(torch.tensor(list(lock.values()), dtype=torch.float32, device=self.kinematics_config.device),
                                        joint_names=list(lock.keys()))

    def get_joint_limits(self) -> "JointLimits":
        k = self.kinematics_config
        return JointLimits(list(self.joint_names), k.joint_limits_position, k.joint_limits_velocity, k.joint_limits_effort)

    def get_self_collision_config(self):
        return self.kinematics_config.self_collision

    def get_active_js(self, full_js):
        """the active (optimised) joints of a joint state that may carry more -- or differently ordered -- joints, selected by
        name (reference ``get_active_js``); a state without names is taken to be in the active order already"""
        names = getattr(full_js, "joint_names", None)
        if not names or list(names) == list(self.joint_names):
            return full_js
        missing = [n for n in self.joint_names if n not in names]
        if missing:
            raise ValueError(f"joint state lacks the active joints {missing}")
        idx = torch.as_tensor([list(names).index(n) for n in self.joint_names], device=full_js.position.device)
        from .types import JointState

        g = lambda t: None if t is None else t.index_select(-1, idx)  # noqa: E731
        return JointState(g(full_js.position), g(full_js.velocity), g(full_js.acceleration), g(full_js.jerk), list(self.joint_names), full_js.dt)

    def get_full_js(self, active_js):
        """active joints + the locked ones at their locked values (zero velocity / acceleration / jerk), reference ``get_full_js``"""
        lock = self.lock_jointstate
        if not lock.joint_names:
            return active_js
        from .types import JointState

        p = active_js.position
        lp = lock.position.to(p.device).expand(*p.shape[:-1], -1)
        cat = lambda t, fill: None if t is None else torch.cat([t, fill], -1)  # noqa: E731
        z = torch.zeros_like(lp)
        return JointState(cat(p, lp), cat(active_js.velocity, z), cat(active_js.acceleration, z), cat(active_js.jerk, z),
                          list(active_js.joint_names or self.joint_names) + list(lock.joint_names), active_js.dt)

    # ---- more of the reference's members (robot/kinematics/kinematics.py:75-100, 278-366, 443-497)
    @property
    def robot_spheres(self) -> torch.Tensor:
        """the collision spheres in their link frames (sphere set 0), [num_spheres, 4]"""
        return self.kinematics_config.link_spheres[0]

    @property
    def all_articulated_joint_names(self) -> List[str]:
        """the active joints followed by the locked ones (reference ``non_fixed_joint_names``, kinematics_loader.py:134)"""
        return list(self.joint_names) + list(self.config.model.lock_joints.keys())

    def get_mimic_js(self, joint_state):
        """state of the mimic joints that follow the actuated joints of ``joint_state`` (reference get_mimic_js, :410-441):
        position = multiplier x actuated + offset; None when the robot has none"""
        from .types import JointState

        mimic = self.config.model.mimic_joints
        if not mimic:
            return None
        names, cols = [], []
        for j, followers in mimic.items():
            q = joint_state.position[..., list(joint_state.joint_names).index(j)]
            for k in followers:
                names.append(k["joint_name"])
                cols.append(k["joint_offset"][0] * q + k["joint_offset"][1])
        if not cols:
            return None
        return JointState.from_position(torch.stack(cols, dim=-1), joint_names=names)

    def update_batch_size(self, batch: int, horizon: int, force_update: bool = False, reset_buffers: bool = False) -> None:
        """allocate the output buffers of ``compute_kinematics`` for [batch, horizon, dof] inputs ahead of the first call"""
        if batch <= 0 or horizon <= 0:
            raise ValueError("batch and horizon must be > 0")
        if reset_buffers:
            self._shape = None
        self._setup(int(batch), int(horizon))

    def get_link_poses(self, joint_position: torch.Tensor, query_link_names: List[str]):
        """poses of tool frames at joint configurations [batch, dof] -> ``Pose`` [batch, len(query_link_names), 3 | 4]
        (``query_link_names`` must be tool frames, as in the reference)"""
        from .types import Pose

        missing = [n for n in query_link_names if n not in self.tool_frames]
        if missing:
            raise ValueError(f"{missing} are not tool frames of this model ({self.tool_frames})")
        q = joint_position if joint_position.ndim == 2 else joint_position.reshape(-1, self.dof)
        tp = self.compute_kinematics(q).tool_poses
        idx = [self.tool_frames.index(n) for n in query_link_names]
        return Pose(tp.position[:, 0][:, idx].contiguous(), tp.quaternion[:, 0][:, idx].contiguous())

    def _link_index(self, link_name: str) -> int:
        names = self.kinematics_config.link_names
        if names is None or link_name not in names:
            raise ValueError(f"link {link_name} is not part of the kinematic model")
        return list(names).index(link_name)

    def get_link_transform(self, link_name: str):
        """fixed offset of a link from its parent joint as a ``Pose`` (the model's ``fixed_transforms`` row)"""
        from .types import Pose

        return Pose.from_matrix(self._fixed_4x4(self.kinematics_config.fixed_transforms[self._link_index(link_name)].unsqueeze(0)))

    def get_all_link_transforms(self):
        from .types import Pose

        return Pose.from_matrix(self._fixed_4x4(self.kinematics_config.fixed_transforms))

    @staticmethod
    def _fixed_4x4(m34: torch.Tensor) -> torch.Tensor:
        m = torch.zeros(m34.shape[0], 4, 4, device=m34.device, dtype=m34.dtype)
        m[:, :3, :4] = m34.reshape(-1, 3, 4)
        m[:, 3, 3] = 1.0
        return m

    def update_kinematics_config(self, new_kin_config) -> None:
        """copy the tensors of another ``KinematicsParams`` of the SAME dimensions into this model's, in place (captured
        graphs and rollouts that hold the tensors see the new values): locked-joint offsets, attached-object spheres ..."""
        import dataclasses

        cur = self.kinematics_config
        for f in dataclasses.fields(cur):
            a, b = getattr(cur, f.name), getattr(new_kin_config, f.name)
            if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
                if a.shape != b.shape:
                    raise ValueError(f"update_kinematics_config: {f.name} changes shape {tuple(a.shape)} -> {tuple(b.shape)}; build a new Kinematics")
                a.copy_(b)

    def get_robot_as_spheres(self, q: torch.Tensor, filter_valid: bool = True):
        """per configuration the robot's collision spheres as ``curobo.scene.Sphere`` objects (reference ``get_robot_as_spheres``);
        ``filter_valid`` drops the disabled ones (radius <= 0)"""
        from .scene.types import Sphere

        sph = self.compute_kinematics(q.reshape(-1, self.dof)).robot_spheres[:, 0].detach().cpu().numpy()
        out = []
        for b in range(sph.shape[0]):
            out.append([Sphere(name=f"robot_sphere_{b}_{i}", pose=[float(x), float(y), float(z), 1, 0, 0, 0], radius=float(r))
                        for i, (x, y, z, r) in enumerate(sph[b]) if not filter_valid or r > 0.0])
        return out

    def _setup(self, b: int, h: int):
        if self._shape != (b, h):
            self._buffers = KinematicsFusedFunction.create_buffers(b, h, self.kinematics_config)
            self._env_idx = torch.zeros(b, dtype=torch.int32, device=self.kinematics_config.device)
            self._shape = (b, h)

    def compute_kinematics(self, joint_state: Union[torch.Tensor, object],
                           idxs_env: Optional[torch.Tensor] = None) -> KinematicsState:
        q = joint_state.position if hasattr(joint_state, "position") else joint_state
        squeeze = q.ndim == 2
        if squeeze:
            q = q.unsqueeze(1)
        b, h, _ = q.shape
        self._setup(b, h)
        bu = self._buffers
        env = self._env_idx if idxs_env is None else idxs_env
        pos, quat, spheres, com, jac = KinematicsFusedFunction.apply(
            q.contiguous(), bu["batch_link_position"], bu["batch_link_quaternion"], bu["batch_robot_spheres"],
            bu["batch_com"], bu["batch_jacobian"], bu["batch_cumul_mat"], self.kinematics_config, bu["grad_out_q"],
            bu["grad_out_q_jacobian"], bu["grad_in_link_pos"], bu["grad_in_link_quat"],
            bu["grad_in_robot_spheres"], bu["grad_in_com"], self.compute_jacobian, self.compute_spheres,
            self.compute_com, env, h)
        return KinematicsState(
            tool_poses=ToolPose(self.tool_frames, pos, quat),
            tool_jacobians=jac if self.compute_jacobian else None,
            robot_spheres=spheres if self.compute_spheres else None,
            robot_com=com if self.compute_com else None)
