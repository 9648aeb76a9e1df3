"""Common data types of the public API -- counterpart of ``curobo.types`` (reference ``curobo/types.py``:
``JointState`` = curobo/_src/state/state_joint.py, ``DeviceCfg`` = _src/types/device_cfg.py, ``Pose`` =
_src/types/pose.py, ``ToolPose`` / ``GoalToolPose`` = _src/types/tool_pose.py).  Only the members the hot-path
front ends use (kinematics, collision checking, IK / trajectory solvers) are mirrored."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import torch


@dataclass
class DeviceCfg:
    """device + dtype of every tensor a module allocates (reference DeviceCfg: default ``cuda:0`` / float32)"""

    device: Union[str, torch.device] = "cuda:0"
    dtype: torch.dtype = torch.float32

    def __post_init__(self):
        self.device = torch.device(self.device)

    def as_torch_dict(self) -> dict:
        return {"device": self.device, "dtype": self.dtype}

    def to_device(self, x) -> torch.Tensor:
        return torch.as_tensor(x, device=self.device, dtype=self.dtype)


@dataclass
class JointState:
    """joint position [..., dof] with optional velocity / acceleration / jerk of the same shape"""

    position: torch.Tensor
    velocity: Optional[torch.Tensor] = None
    acceleration: Optional[torch.Tensor] = None
    jerk: Optional[torch.Tensor] = None
    joint_names: Optional[List[str]] = None
    dt: Optional[torch.Tensor] = None

    @staticmethod
    def from_position(position, joint_names: Optional[List[str]] = None) -> "JointState":
        p = position if torch.is_tensor(position) else torch.as_tensor(position, dtype=torch.float32)
        return JointState(position=p, joint_names=list(joint_names) if joint_names is not None else None)

    @staticmethod
    def zeros(shape: Sequence[int], device_cfg: Optional[DeviceCfg] = None, joint_names: Optional[List[str]] = None) -> "JointState":
        kw = (device_cfg or DeviceCfg()).as_torch_dict()
        z = lambda: torch.zeros(*shape, **kw)  # noqa: E731
        return JointState(z(), z(), z(), z(), joint_names)

    @property
    def shape(self):
        return self.position.shape

    def clone(self) -> "JointState":
        c = lambda t: None if t is None else t.clone()  # noqa: E731
        return JointState(c(self.position), c(self.velocity), c(self.acceleration), c(self.jerk),
                          None if self.joint_names is None else list(self.joint_names), c(self.dt))

    def detach(self) -> "JointState":
        d = lambda t: None if t is None else t.detach()  # noqa: E731
        return JointState(d(self.position), d(self.velocity), d(self.acceleration), d(self.jerk), self.joint_names, d(self.dt))

    def __getitem__(self, idx) -> "JointState":
        g = lambda t: None if t is None else t[idx]  # noqa: E731
        return JointState(g(self.position), g(self.velocity), g(self.acceleration), g(self.jerk), self.joint_names, self.dt)

    def __len__(self) -> int:
        return int(self.position.shape[0])


@dataclass
class Pose:
    """position [..., 3] and quaternion [..., 4] (w, x, y, z)"""

    position: torch.Tensor
    quaternion: torch.Tensor

    def __post_init__(self):
        if not torch.is_tensor(self.position):
            self.position = torch.as_tensor(self.position, dtype=torch.float32)
        if not torch.is_tensor(self.quaternion):
            self.quaternion = torch.as_tensor(self.quaternion, dtype=torch.float32)


@dataclass
class GoalToolPose:
    """goal poses per tool frame: position [batch, T, num_goalset, 3], quaternion [batch, T, num_goalset, 4] (wxyz)
    (reference GoalToolPose, _src/types/tool_pose.py)"""

    tool_frames: List[str]
    position: torch.Tensor
    quaternion: torch.Tensor

    @property
    def batch_size(self) -> int:
        return int(self.position.shape[0])

    @property
    def num_goalset(self) -> int:
        return int(self.position.shape[2])


@dataclass
class ToolPoseCriteria:
    """per-axis weights of the pose cost (reference _src/cost/tool_pose_criteria.py): terminal / non-terminal
    position + rotation axis factors"""

    terminal_pose_axes_weight_factor: List[float] = field(default_factory=lambda: [1.0] * 6)
    non_terminal_pose_axes_weight_factor: List[float] = field(default_factory=lambda: [0.0] * 6)
