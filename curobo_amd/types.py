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
        # (zero velocity / acceleration / jerk, as the reference's from_position: state_joint.py:133-140)
        return JointState(p, p * 0.0, p * 0.0, p * 0.0, list(joint_names) if joint_names is not None else None)

    @staticmethod
    def zeros(shape: Sequence[int], device_cfg: Optional[DeviceCfg] = None, joint_names: Optional[List[str]] = None) -> "JointState":
        kw = (device_cfg or DeviceCfg()).as_torch_dict()
        z = lambda: torch.zeros(*shape, **kw)  # noqa: E731
        return JointState(z(), z(), z(), z(), joint_names, torch.ones(shape[0], **kw))  # (dt of ones per row: reference :159-169)

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
        if isinstance(idx, list):
            idx = torch.as_tensor(idx, device=self.position.device, dtype=torch.long)
        g = lambda t: None if t is None else t[idx]  # noqa: E731
        dt = self.dt
        if dt is not None and dt.ndim > 0 and dt.shape[0] > 1:  # a per-row dt follows the rows (kept 1-d for one row)
            if isinstance(idx, int) or (torch.is_tensor(idx) and idx.numel() == 1):
                i = int(idx)
                dt = dt[i:i + 1]
            elif isinstance(idx, slice) or torch.is_tensor(idx):
                dt = dt[idx]
        return JointState(g(self.position), g(self.velocity), g(self.acceleration), g(self.jerk), self.joint_names, dt)

    def __len__(self) -> int:
        return int(self.position.shape[0])

    # ------------------------------------------------------------------ the reference's members (state/state_joint.py,
    # state_joint_ops.py, state_joint_trajectory_ops.py), pure torch; held to the reference's class on random states by
    # tests/golden/compare_joint_state.py
    def _fields(self):
        return (self.position, self.velocity, self.acceleration, self.jerk)

    def _like(self, p, v, a, j, joint_names="same", dt="same") -> "JointState":
        return JointState(p, v, a, j, self.joint_names if isinstance(joint_names, str) else joint_names, self.dt if isinstance(dt, str) else dt)

    def _map(self, fn, dt="same") -> "JointState":
        m = lambda t: None if t is None else fn(t)  # noqa: E731
        return self._like(m(self.position), m(self.velocity), m(self.acceleration), m(self.jerk), dt=dt)

    @property
    def device(self) -> torch.device:
        return self.position.device

    @property
    def dtype(self) -> torch.dtype:
        return self.position.dtype

    @property
    def ndim(self) -> int:
        return self.position.ndim

    def data_ptr(self) -> int:
        return self.position.data_ptr()

    @staticmethod
    def from_numpy(joint_names: List[str], position, velocity=None, acceleration=None, jerk=None,
                   device_cfg: Optional[DeviceCfg] = None) -> "JointState":
        cfg = device_cfg or DeviceCfg()
        pos = cfg.to_device(position)
        f = lambda x: cfg.to_device(x) if x is not None else pos * 0.0  # noqa: E731
        return JointState(pos, f(velocity), f(acceleration), f(jerk), list(joint_names) if joint_names is not None else None)

    @staticmethod
    def from_state_tensor(state_tensor: torch.Tensor, joint_names: Optional[List[str]] = None, dof: int = 7) -> "JointState":
        c = lambda k: state_tensor[..., k * dof:(k + 1) * dof].contiguous()  # noqa: E731
        return JointState(c(0), c(1), c(2), c(3), joint_names)

    @staticmethod
    def from_list(position, velocity, acceleration, device_cfg: Optional[DeviceCfg] = None) -> "JointState":
        cfg = device_cfg or DeviceCfg()
        return JointState(cfg.to_device(position), cfg.to_device(velocity), cfg.to_device(acceleration))

    def to(self, device_cfg) -> "JointState":
        """a ``DeviceCfg`` (the reference's argument) or a torch device"""
        kw = device_cfg.as_torch_dict() if isinstance(device_cfg, DeviceCfg) else {"device": device_cfg}
        return self._map(lambda t: t.to(**kw), dt=None if self.dt is None else self.dt.to(**kw))

    def copy_reference(self, in_joint_state: "JointState") -> "JointState":
        for f in ("position", "velocity", "acceleration", "jerk", "dt", "joint_names"):
            setattr(self, f, getattr(in_joint_state, f))
        return self

    def _same_shape(self, other: "JointState") -> bool:
        for mine, theirs in zip(self._fields()[:3], other._fields()[:3]):
            if theirs is not None and (mine is None or mine.shape != theirs.shape):
                return False
        return True

    def copy_(self, in_joint_state: "JointState", allow_clone: bool = True) -> "JointState":
        """in place when the shapes agree; else (``allow_clone``) this state takes clones of the other's tensors"""
        if in_joint_state.joint_names is not None:
            self.joint_names = in_joint_state.joint_names
        if self._same_shape(in_joint_state):
            for f in ("position", "velocity", "acceleration", "jerk", "dt"):
                mine, theirs = getattr(self, f), getattr(in_joint_state, f)
                if mine is not None and theirs is not None:
                    mine.copy_(theirs)
            return self
        if not allow_clone:
            raise ValueError(f"current state has shape: {tuple(self.position.shape)} while new shape is {tuple(in_joint_state.position.shape)}")
        return self.copy_reference(in_joint_state.clone())

    def copy_data(self, in_joint_state: "JointState") -> "JointState":
        return self.copy_(in_joint_state)

    def unsqueeze(self, idx: int) -> "JointState":
        return self._map(lambda t: t.unsqueeze(idx))

    def squeeze(self, dim: Optional[int] = 0) -> "JointState":
        return self._map(lambda t: torch.squeeze(t, dim))

    def view(self, *shape) -> "JointState":
        dt = self.dt.view(*shape[:2]) if len(shape) > 2 and self.dt is not None else self.dt
        return self._map(lambda t: t.view(*shape), dt=dt)

    def __setitem__(self, idx, value: "JointState") -> None:
        for f in ("position", "velocity", "acceleration", "jerk"):
            getattr(self, f)[idx] = getattr(value, f)
        if self.dt is not None:
            self.dt[idx] = value.dt

    def get_state_tensor(self) -> torch.Tensor:
        z = lambda t: self.position * 0.0 if t is None else t  # noqa: E731
        return torch.cat((self.position, z(self.velocity), z(self.acceleration), z(self.jerk)), dim=-1)

    def stack(self, new_state: "JointState") -> "JointState":
        return JointState.from_state_tensor(torch.cat((self.get_state_tensor(), new_state.get_state_tensor()), dim=-2),
                                            joint_names=self.joint_names, dof=self.position.shape[-1])

    def cat(self, other_js: "JointState", dim: int) -> "JointState":
        c = lambda a, b: torch.cat((a, b), dim=dim) if a is not None and b is not None else None  # noqa: E731
        names = (self.joint_names + other_js.joint_names) if dim == -1 else self.joint_names
        return JointState(*(c(a, b) for a, b in zip(self._fields(), other_js._fields())), names, self.dt)

    def repeat(self, repeat_input: List[int]) -> "JointState":
        return self._map(lambda t: t.repeat(repeat_input))

    def repeat_seeds(self, num_seeds: int) -> "JointState":
        r = lambda t: t.view(t.shape[0], 1, t.shape[-1]).repeat(1, num_seeds, 1).reshape(t.shape[0] * num_seeds, t.shape[-1])  # noqa: E731
        return self._map(r, dt=None if self.dt is None else r(self.dt))

    def apply_kernel(self, kernel_mat: torch.Tensor) -> "JointState":
        return JointState(kernel_mat @ self.position, kernel_mat @ self.velocity, kernel_mat @ self.acceleration,
                          kernel_mat @ self.jerk if self.jerk is not None else None, self.joint_names,
                          kernel_mat @ self.dt if self.dt is not None else None)

    def blend(self, coeff, new_state: "JointState") -> "JointState":
        """in place: x <- c x_new + (1 - c) x with the coefficients coeff.position / .velocity / .acceleration / .jerk"""
        for f in ("position", "velocity", "acceleration", "jerk"):
            c = getattr(coeff, f)
            getattr(self, f)[:] = c * getattr(new_state, f) + (1.0 - c) * getattr(self, f)
        return self

    def scale(self, dt) -> "JointState":
        m = lambda t, k: None if t is None else t * (dt ** k)  # noqa: E731
        return JointState(self.position, m(self.velocity, 1), m(self.acceleration, 2), m(self.jerk, 3), self.joint_names)

    def scale_by_dt(self, dt: torch.Tensor, new_dt: torch.Tensor) -> "JointState":
        s = dt / new_dt
        if self.velocity is not None and self.velocity.ndim in (2, 3):
            s = s.view(-1, *([1] * (self.velocity.ndim - 1)))
        v = None if self.velocity is None else self.velocity * s
        a = None if self.acceleration is None else self.acceleration * s * s
        j = None if self.jerk is None else self.jerk * s * s * s
        return JointState(self.position, v, a, j, self.joint_names, new_dt)

    def scale_time(self, new_dt: torch.Tensor) -> "JointState":
        return self.scale_by_dt(self.dt, new_dt)

    def calculate_fd_from_position(self, dt: Optional[torch.Tensor] = None) -> "JointState":
        """velocity, acceleration, jerk by forward differences along the horizon (each one sample shorter), in place"""
        if self.dt is None and dt is None:
            raise ValueError("dt is required")
        dt = self.dt if dt is None else dt
        fd = lambda p: ((torch.roll(p, -1, -2) - p) * (1 / dt).unsqueeze(-1))[..., :-1, :]  # noqa: E731
        self.velocity = fd(self.position)
        self.acceleration = fd(self.velocity)
        self.jerk = fd(self.acceleration)
        return self

    def reindex(self, joint_names: List[str]) -> None:
        """in place: the joints in the order of ``joint_names`` (a subset is allowed)"""
        if self.joint_names is None:
            raise ValueError("joint names are not specified in JointState")
        idx = [self.joint_names.index(j) for j in joint_names]
        self.joint_names = [self.joint_names[i] for i in idx]
        sel = torch.as_tensor(idx, device=self.position.device, dtype=torch.long)
        for f in ("position", "velocity", "acceleration", "jerk"):
            t = getattr(self, f)
            if t is not None:
                setattr(self, f, torch.index_select(t, -1, sel))

    def reorder(self, joint_names: List[str]) -> "JointState":
        """a new state with the joints in the order of ``joint_names``"""
        out = self.clone()
        out.reindex(joint_names)
        return out

    def get_ordered_joint_state(self, ordered_joint_names: List[str]) -> "JointState":
        return self.reorder(ordered_joint_names)

    def get_augmented_joint_state(self, joint_names: List[str], lock_joints: Optional["JointState"] = None) -> "JointState":
        if lock_joints is None:
            return self.reorder(joint_names)
        if joint_names is None or self.joint_names is None:
            raise ValueError("joint_names can't be None")
        if any(n in self.joint_names for n in lock_joints.joint_names):
            raise ValueError("lock_joints is also listed in js.joint_names")
        return self.clone().append_joints(lock_joints).reorder(joint_names)

    def append_joints(self, joint_state: "JointState") -> "JointState":
        """the joints of ``joint_state`` (one configuration, or one per row) behind this state's, zero derivatives for them"""
        other = joint_state
        if not other.joint_names:
            raise ValueError("joint_names are required to append")
        cur = self if self.position.ndim > 1 else self.unsqueeze(0)
        lead = cur.position.shape[:-1]
        op = other.position.reshape(-1, other.position.shape[-1])
        if op.shape[0] not in (1, int(torch.tensor(lead).prod())):
            raise ValueError("appending joints requires the new joints to have a shape matching current batch size or have a batch size of 1.")
        extra = op.expand(int(torch.tensor(lead).prod()), -1).reshape(*lead, -1).to(cur.position.dtype)
        z = torch.zeros_like(extra)
        c = lambda t, e: None if t is None else torch.cat((t, e), dim=-1)  # noqa: E731
        out = JointState(c(cur.position, extra), c(cur.velocity, z), c(cur.acceleration, z), c(cur.jerk, z),
                         list(self.joint_names) + list(other.joint_names), other.dt if other.dt is not None else self.dt)
        return out if self.position.ndim > 1 else out.squeeze(0)

    def gather_by_seed_index(self, idx: torch.Tensor) -> "JointState":
        """[batch, seeds, horizon, dof] state, idx [batch, k] -> the k chosen seeds of every problem [batch, k, horizon, dof]"""
        if idx.ndim != 2 or idx.shape[0] != self.position.shape[0] or self.position.ndim != 4:
            raise ValueError("gather_by_seed_index: idx [batch, k] against a [batch, seeds, horizon, dof] state")
        B, S, H, D = self.position.shape
        flat = (idx + torch.arange(B, device=idx.device).view(-1, 1) * S).view(-1)
        g = lambda t: None if t is None else t.reshape(B * S, H, D)[flat].view(B, idx.shape[1], H, D)  # noqa: E731
        dt = None if self.dt is None else self.dt.reshape(B * S, -1)[flat].view(B, idx.shape[1])
        return JointState(g(self.position), g(self.velocity), g(self.acceleration), g(self.jerk), self.joint_names, dt)

    def copy_only_index(self, in_joint_state: "JointState", idx) -> "JointState":
        for f in ("position", "velocity", "acceleration", "jerk", "dt"):
            if getattr(self, f) is not None:
                getattr(self, f)[idx] = getattr(in_joint_state, f)[idx]
        return self

    def copy_at_index(self, in_joint_state: "JointState", idx) -> None:
        top = idx if isinstance(idx, int) else (max(idx) if isinstance(idx, list) else int(torch.max(idx)))
        if top >= self.position.shape[0]:
            raise ValueError(f"{top} index out of range, current state is of length {self.position.shape[0]}")
        for f in ("position", "velocity", "acceleration", "jerk"):
            if getattr(self, f) is not None:
                getattr(self, f)[idx] = getattr(in_joint_state, f)
        if self.dt is not None and in_joint_state.dt is not None:
            self.dt[idx] = in_joint_state.dt

    def copy_at_batch_seed_indices(self, in_joint_state: "JointState", batch_idx: torch.Tensor, seed_idx: torch.Tensor) -> "JointState":
        for f in ("position", "velocity", "acceleration", "jerk", "dt"):
            if getattr(self, f) is not None and getattr(in_joint_state, f) is not None:
                getattr(self, f)[batch_idx, seed_idx] = getattr(in_joint_state, f)[batch_idx, seed_idx]
        return self

    def get_trajectory_at_horizon_index(self, horizon_index: int) -> "JointState":
        if self.position.ndim < 2:
            raise ValueError("JointState does not have horizon")
        return self._map(lambda t: t[..., horizon_index, :])

    def trim_trajectory(self, start_idx: int, end_idx: Optional[int] = None) -> "JointState":
        if self.position.ndim < 2:
            raise ValueError("JointState does not have horizon")
        end = self.position.shape[-2] if not end_idx else end_idx
        return self._map(lambda t: t[..., start_idx:end, :])

    def index_dof(self, idx: torch.Tensor) -> "JointState":
        names = [self.joint_names[int(i)] for i in idx]
        return JointState(*(None if t is None else torch.index_select(t, -1, idx) for t in self._fields()), names)


def _quat_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)


def _quat_rotate(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    w, u = q[..., :1], q[..., 1:]
    t = 2.0 * torch.cross(u, v, dim=-1)
    return v + w * t + torch.cross(u, t, dim=-1)


@dataclass
class Pose:
    """position [..., 3] and quaternion [..., 4] (w, x, y, z) (reference _src/types/pose.py: the members the planners use)"""

    position: torch.Tensor
    quaternion: torch.Tensor
    name: Optional[str] = None

    def __post_init__(self):
        if not torch.is_tensor(self.position):
            self.position = torch.as_tensor(self.position, dtype=torch.float32)
        if not torch.is_tensor(self.quaternion):
            self.quaternion = torch.as_tensor(self.quaternion, dtype=torch.float32)

    @staticmethod
    def from_list(pose: Sequence[float], device_cfg: Optional[DeviceCfg] = None) -> "Pose":
        """[x, y, z, qw, qx, qy, qz] -> Pose with a batch of one (reference Pose.from_list)"""
        dev = device_cfg.device if device_cfg is not None else None
        t = torch.as_tensor([float(x) for x in pose], dtype=torch.float32, device=dev)
        return Pose(t[:3].view(1, 3).clone(), t[3:7].view(1, 4).clone())

    @staticmethod
    def from_matrix(matrix: torch.Tensor) -> "Pose":
        """homogeneous transforms [..., 4, 4] (or [..., 3, 4]) -> Pose, quaternion wxyz with w >= 0 (reference Pose.from_matrix;
        the rotation through the largest of the four squared components, the usual branch-stable conversion)"""
        m = matrix.reshape(-1, matrix.shape[-2], matrix.shape[-1]).to(torch.float32)
        R, t = m[:, :3, :3], m[:, :3, 3]
        m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
        four = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1).clamp_min(0.0)
        k = four.argmax(-1)
        h = 0.5 * torch.sqrt(four.gather(-1, k.unsqueeze(-1)).squeeze(-1)).clamp_min(1e-12)
        q4 = 0.25 / h
        cand = torch.stack([
            torch.stack([h, (R[:, 2, 1] - R[:, 1, 2]) * q4, (R[:, 0, 2] - R[:, 2, 0]) * q4, (R[:, 1, 0] - R[:, 0, 1]) * q4], -1),
            torch.stack([(R[:, 2, 1] - R[:, 1, 2]) * q4, h, (R[:, 0, 1] + R[:, 1, 0]) * q4, (R[:, 0, 2] + R[:, 2, 0]) * q4], -1),
            torch.stack([(R[:, 0, 2] - R[:, 2, 0]) * q4, (R[:, 0, 1] + R[:, 1, 0]) * q4, h, (R[:, 1, 2] + R[:, 2, 1]) * q4], -1),
            torch.stack([(R[:, 1, 0] - R[:, 0, 1]) * q4, (R[:, 0, 2] + R[:, 2, 0]) * q4, (R[:, 1, 2] + R[:, 2, 1]) * q4, h], -1)], 1)
        q = cand[torch.arange(m.shape[0], device=m.device), k]
        q = torch.where(q[:, :1] < 0, -q, q)
        lead = matrix.shape[:-2]
        return Pose(t.reshape(*lead, 3).contiguous(), q.reshape(*lead, 4).contiguous())

    def to(self, device=None, device_cfg: Optional[DeviceCfg] = None) -> "Pose":
        """a device, or (as the reference's first argument) a ``DeviceCfg``"""
        if device_cfg is None and isinstance(device, DeviceCfg):
            device_cfg, device = device, None
        if device_cfg is not None:
            device = device_cfg.device
        if device is None:
            raise ValueError("Pose.to() requires device_cfg or device")
        return Pose(self.position.to(device), self.quaternion.to(device), self.name)

    def clone(self) -> "Pose":
        return Pose(self.position.clone(), self.quaternion.clone(), self.name)

    def multiply(self, other: "Pose") -> "Pose":
        """self * other: ``other`` is expressed in the frame of ``self`` (reference Pose.multiply)"""
        q1, p1 = self.quaternion, self.position
        q2, p2 = other.quaternion.to(q1.device), other.position.to(p1.device)
        return Pose(p1 + _quat_rotate(q1, p2.expand_as(p1) if p2.shape[0] == 1 else p2), _quat_mul(q1, q2))

    def inverse(self) -> "Pose":
        qi = self.quaternion * self.quaternion.new_tensor([1.0, -1.0, -1.0, -1.0])
        return Pose(-_quat_rotate(qi, self.position), qi)

    # ------------------------------------------------------------------ the reference's convenience members (types/pose.py:117-670),
    # pure torch; held to the reference's class on random poses by tests/golden/compare_pose_ops.py
    @property
    def device(self):
        return self.position.device

    @property
    def ndim(self) -> int:
        return self.position.ndim

    @property
    def shape(self):
        return self.position.shape

    @property
    def batch_size(self) -> int:
        return int(self.position.shape[0]) if self.position.ndim > 1 else 1

    batch = batch_size  # (deprecated name in the reference)

    def __len__(self) -> int:
        return self.batch_size

    def __getitem__(self, idx) -> "Pose":
        return Pose(self.position[idx], self.quaternion[idx])

    def __setitem__(self, idx, value: "Pose") -> None:
        self.position[idx] = value.position
        self.quaternion[idx] = value.quaternion

    def get_index(self, b: int, n: Optional[int] = None) -> "Pose":
        return Pose(self.position[b, :], self.quaternion[b, :]) if n is None else Pose(self.position[b, n, :], self.quaternion[b, n, :])

    def detach(self) -> "Pose":
        self.position, self.quaternion = self.position.detach(), self.quaternion.detach()
        return self

    def requires_grad_(self, requires_grad: bool) -> None:
        self.position.requires_grad_(requires_grad)
        self.quaternion.requires_grad_(requires_grad)

    def contiguous(self) -> "Pose":
        return Pose(self.position.contiguous(), self.quaternion.contiguous(), self.name)

    def copy_(self, pose: "Pose") -> None:
        if pose.position.shape != self.position.shape or pose.quaternion.shape != self.quaternion.shape:
            raise ValueError(f"Copy not possible due to shape mismatch: {tuple(pose.position.shape)} != {tuple(self.position.shape)}")
        self.position.copy_(pose.position)
        self.quaternion.copy_(pose.quaternion)

    @staticmethod
    def from_numpy(position, quaternion, device_cfg: Optional[DeviceCfg] = None) -> "Pose":
        dev = device_cfg.device if device_cfg is not None else None
        return Pose(torch.as_tensor(position, dtype=torch.float32, device=dev), torch.as_tensor(quaternion, dtype=torch.float32, device=dev))

    @staticmethod
    def from_batch_list(pose: Sequence[Sequence[float]], device_cfg: Optional[DeviceCfg] = None, q_xyzw: bool = False) -> "Pose":
        dev = device_cfg.device if device_cfg is not None else None
        m = torch.as_tensor(pose, dtype=torch.float32, device=dev)
        q = m[..., 3:7]
        if q_xyzw:
            q = q[..., [3, 0, 1, 2]]
        return Pose(m[..., :3].contiguous(), q.contiguous())

    @staticmethod
    def _euler_to_quaternion(euler_xyz: torch.Tensor, intrinsic: bool) -> torch.Tensor:
        h = euler_xyz * 0.5
        cx, cy, cz = torch.cos(h[..., 0:1]), torch.cos(h[..., 1:2]), torch.cos(h[..., 2:3])
        sx, sy, sz = torch.sin(h[..., 0:1]), torch.sin(h[..., 1:2]), torch.sin(h[..., 2:3])
        s = -1.0 if intrinsic else 1.0  # extrinsic XYZ: q = qz qy qx; intrinsic XYZ: q = qx qy qz
        return torch.cat([cx * cy * cz + s * sx * sy * sz, sx * cy * cz - s * cx * sy * sz, cx * sy * cz + s * sx * cy * sz,
                          cx * cy * sz - s * sx * sy * cz], dim=-1)

    @classmethod
    def _from_euler(cls, euler_xyz: torch.Tensor, position: Optional[torch.Tensor], intrinsic: bool) -> "Pose":
        q = cls._euler_to_quaternion(euler_xyz, intrinsic)
        if q.ndim == 1:
            q = q.unsqueeze(0)
        if position is None:
            position = torch.zeros(q.shape[:-1] + (3,), device=euler_xyz.device, dtype=euler_xyz.dtype)
        elif position.ndim == 1:
            position = position.unsqueeze(0)
        return cls(position, q)

    @classmethod
    def from_euler_xyz(cls, euler_xyz: torch.Tensor, position: Optional[torch.Tensor] = None) -> "Pose":
        """rotations about the FIXED world axes X, then Y, then Z (reference from_euler_xyz)"""
        return cls._from_euler(euler_xyz, position, False)

    @classmethod
    def from_euler_xyz_intrinsic(cls, euler_xyz: torch.Tensor, position: Optional[torch.Tensor] = None) -> "Pose":
        """rotations about the BODY axes X, then Y, then Z, as a chain of revolute joints composes them"""
        return cls._from_euler(euler_xyz, position, True)

    def get_rotation(self) -> torch.Tensor:
        """rotation matrices [..., 3, 3]"""
        w, x, y, z = self.quaternion.unbind(-1)
        return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1).reshape(*self.quaternion.shape[:-1], 3, 3)

    get_rotation_matrix = get_rotation

    def get_affine_matrix(self, out_matrix: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[..., 3, 4] = (R | t)"""
        m = torch.cat([self.get_rotation(), self.position.unsqueeze(-1)], dim=-1)
        if out_matrix is not None:
            out_matrix.copy_(m)
            return out_matrix
        return m

    def get_matrix(self, out_matrix: Optional[torch.Tensor] = None) -> torch.Tensor:
        """homogeneous transforms [..., 4, 4]"""
        m = torch.zeros(*self.position.shape[:-1], 4, 4, device=self.position.device, dtype=self.position.dtype)
        m[..., :3, :4] = self.get_affine_matrix()
        m[..., 3, 3] = 1.0
        if out_matrix is not None:
            out_matrix.copy_(m)
            return out_matrix
        return m

    def get_numpy_matrix(self):
        return self.get_matrix().cpu().numpy()

    def get_numpy_affine_matrix(self):
        return self.get_affine_matrix().cpu().numpy()

    def get_pose_vector(self) -> torch.Tensor:
        return torch.cat((self.position, self.quaternion), dim=-1)

    def tolist(self, q_xyzw: bool = False) -> List[float]:
        q = self.quaternion.cpu().squeeze().tolist()
        return self.position.cpu().squeeze().tolist() + ([q[1], q[2], q[3], q[0]] if q_xyzw else q)

    to_list = tolist

    def stack(self, other_pose: "Pose") -> "Pose":
        return Pose(torch.vstack((self.position, other_pose.position)), torch.vstack((self.quaternion, other_pose.quaternion)))

    @staticmethod
    def cat(pose_list: List["Pose"]) -> "Pose":
        return Pose(torch.cat([p.position for p in pose_list]), torch.cat([p.quaternion for p in pose_list]))

    def repeat(self, n: int) -> "Pose":
        return self if n <= 1 else Pose(self.position.repeat(n, 1), self.quaternion.repeat(n, 1))

    def repeat_seeds(self, num_seeds: int) -> "Pose":
        if num_seeds <= 1:
            return Pose(self.position, self.quaternion)
        b = self.batch_size
        return Pose(self.position.view(b, 1, 3).repeat(1, num_seeds, 1).reshape(b * num_seeds, 3),
                    self.quaternion.view(b, 1, 4).repeat(1, num_seeds, 1).reshape(b * num_seeds, 4))

    def unsqueeze(self, dim: int = -1) -> "Pose":
        self.position, self.quaternion = self.position.unsqueeze(dim), self.quaternion.unsqueeze(dim)
        return self

    def squeeze(self, dim: int = -1) -> "Pose":
        self.position, self.quaternion = self.position.squeeze(dim), self.quaternion.squeeze(dim)
        return self

    def apply_kernel(self, kernel_mat: torch.Tensor) -> "Pose":
        return Pose(kernel_mat @ self.position, kernel_mat @ self.quaternion)

    def linear_distance(self, other_pose: "Pose") -> torch.Tensor:
        return torch.linalg.norm(self.position - other_pose.position, dim=-1)

    def angular_distance(self, other_pose: "Pose", use_phi3: bool = False) -> torch.Tensor:
        """angle of the relative rotation in radians, or with ``use_phi3`` Huynh's phi_3 in [0, 1] (reference geom/quaternion.py:75-135)"""
        a = self.quaternion / self.quaternion.norm(dim=-1, keepdim=True)
        b = other_pose.quaternion / other_pose.quaternion.norm(dim=-1, keepdim=True)
        if use_phi3:
            return torch.acos(torch.clamp((a * b).sum(-1).abs(), 0.0, 1.0)) / (torch.pi * 0.5)
        rel = _quat_mul(a, b * b.new_tensor([1.0, -1.0, -1.0, -1.0]))
        return 2.0 * torch.atan2(rel[..., 1:].norm(dim=-1), rel[..., 0].abs())

    def distance(self, other_pose: "Pose", use_phi3: bool = False):
        return self.linear_distance(other_pose), self.angular_distance(other_pose, use_phi3)

    def transform_points(self, points: torch.Tensor, *unused) -> torch.Tensor:
        """ONE pose applied to points [n, 3] (any leading shape is flattened) -> [n, 3]"""
        pts = points.reshape(-1, 3)
        return self.position.reshape(-1, 3)[:1] + _quat_rotate(self.quaternion.reshape(-1, 4)[:1].expand(pts.shape[0], 4), pts)

    transform_point = transform_points  # (deprecated name in the reference)

    def batch_transform_points(self, points: torch.Tensor, *unused) -> torch.Tensor:
        """pose b applied to points [b, n, 3] -> [b, n, 3]"""
        if points.ndim <= 2:
            raise ValueError("batch_transform requires points to be b,n,3 shape")
        p, q = self.position.reshape(-1, 1, 3), self.quaternion.reshape(-1, 1, 4)
        return p + _quat_rotate(q.expand(points.shape[0], points.shape[1], 4), points)

    def batch_transform_points_inverse(self, points: torch.Tensor, *unused) -> torch.Tensor:
        return self.inverse().batch_transform_points(points)

    def compute_offset_pose(self, offset: "Pose") -> "Pose":
        return self.multiply(offset)

    def compute_local_pose(self, world_pose: "Pose") -> "Pose":
        return self.inverse().multiply(world_pose)


class _FramePoseMembers:
    """what the reference's ``ToolPose`` [batch, horizon, links, 3 | 4] and ``GoalToolPose`` [batch, horizon, links, goal set, 3 | 4]
    share (types/tool_pose.py:54-163, 214-357): the per-frame accessors and the tensor-wise copies, on ``tool_frames`` /
    ``position`` / ``quaternion``"""

    @property
    def ndim(self) -> int:
        return self.position.ndim

    def get_link_pose(self, link_name: str, make_contiguous: bool = False) -> "Pose":
        """one frame's poses flattened to [batch * horizon (* goal set), 3 | 4]"""
        if link_name not in self.tool_frames:
            raise ValueError(f"Link {link_name} not found in {self.tool_frames}")
        li = self.tool_frames.index(link_name)
        pos, quat = self.position[:, :, li].reshape(-1, 3), self.quaternion[:, :, li].reshape(-1, 4)
        if make_contiguous:
            pos, quat = pos.contiguous(), quat.contiguous()
        return Pose(pos, quat, link_name)

    def to_dict(self, make_contiguous: bool = True):
        return {n: self.get_link_pose(n, make_contiguous) for n in self.tool_frames}

    def copy_(self, other) -> None:
        self.tool_frames = other.tool_frames
        self.position.copy_(other.position)
        self.quaternion.copy_(other.quaternion)

    def requires_grad_(self, requires_grad: bool) -> None:
        self.position.requires_grad_(requires_grad)
        self.quaternion.requires_grad_(requires_grad)

    def clone(self):
        return type(self)(list(self.tool_frames), self.position.clone(), self.quaternion.clone())

    def detach(self):
        return type(self)(list(self.tool_frames), self.position.detach(), self.quaternion.detach())

    def contiguous(self):
        return type(self)(self.tool_frames, self.position.contiguous(), self.quaternion.contiguous())

    def __len__(self) -> int:
        return len(self.tool_frames)

    def __getitem__(self, idx):
        """a frame name -> its ``Pose``; an int -> that batch entry (batch axis kept); anything else indexes the batch axis"""
        if isinstance(idx, str):
            return self.get_link_pose(idx)
        if isinstance(idx, int):
            return type(self)(self.tool_frames, self.position[idx].unsqueeze(0), self.quaternion[idx].unsqueeze(0))
        return type(self)(self.tool_frames, self.position[idx], self.quaternion[idx])

    def reorder_links(self, ordered_tool_frames: List[str]):
        """the frames in another order, or a subset of them"""
        if not set(ordered_tool_frames).issubset(self.tool_frames):
            raise ValueError(f"Ordered link names {ordered_tool_frames} not a subset of {self.tool_frames}")
        if list(self.tool_frames) == list(ordered_tool_frames):
            return self
        idx = [self.tool_frames.index(n) for n in ordered_tool_frames]
        return type(self)(list(ordered_tool_frames), self.position[:, :, idx].contiguous(), self.quaternion[:, :, idx].contiguous())


@dataclass
class GoalToolPose(_FramePoseMembers):
    """goal poses per tool frame (reference GoalToolPose, _src/types/tool_pose.py:182-340): position [batch, horizon,
    num_links, num_goalset, 3], quaternion [batch, horizon, num_links, num_goalset, 4] (wxyz).  The solvers of this package
    take static goals: with horizon > 1 they read the last entry."""

    tool_frames: List[str]
    position: torch.Tensor
    quaternion: torch.Tensor

    def __post_init__(self):
        if self.position.ndim != 5:
            raise ValueError(f"GoalToolPose position must be 5D [B,H,L,G,3], got {tuple(self.position.shape)}")
        if self.quaternion.ndim != 5:
            raise ValueError(f"GoalToolPose quaternion must be 5D [B,H,L,G,4], got {tuple(self.quaternion.shape)}")
        if self.position.shape[2] != len(self.tool_frames):
            raise ValueError(f"num_links dim ({self.position.shape[2]}) != len(tool_frames) ({len(self.tool_frames)})")

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
    def num_goalset(self) -> int:
        return int(self.position.shape[3])

    @property
    def shape(self):
        return self.position.shape

    @property
    def device(self):
        return self.position.device

    def static_goals(self, ordered_tool_frames: Optional[List[str]] = None):
        """(position [batch, num_links, num_goalset, 3], quaternion [batch, num_links, num_goalset, 4]): what the solvers read.
        ``ordered_tool_frames`` = the robot's tool frames: the goal must name exactly those (any order) and is returned in
        that order -- the kernels index goals by the robot's frame order (reference ToolPose.reorder_links,
        tool_pose.py:147-163)."""
        pos, quat = self.position[:, -1], self.quaternion[:, -1]
        if ordered_tool_frames is None or list(ordered_tool_frames) == list(self.tool_frames):
            return pos, quat
        want = list(ordered_tool_frames)
        if len(want) != len(self.tool_frames) or set(want) != set(self.tool_frames):
            raise ValueError(f"the goal names the tool frames {list(self.tool_frames)}, the robot has {want}: every tool frame needs a goal "
                             "(ToolPoseCriteria can switch a frame's axes off)")
        idx = torch.as_tensor([self.tool_frames.index(n) for n in want], device=pos.device)
        return pos.index_select(1, idx), quat.index_select(1, idx)

    @classmethod
    def from_poses(cls, pose_dict, ordered_tool_frames: Optional[List[str]] = None, num_goalset: int = 1) -> "GoalToolPose":
        """per-frame ``Pose`` objects with position [batch * num_goalset, 3] -> GoalToolPose [batch, 1, num_links,
        num_goalset, 3 | 4] (reference from_poses, :242-277)"""
        if not pose_dict:
            raise ValueError("pose_dict cannot be empty")
        frames = list(ordered_tool_frames) if ordered_tool_frames else list(pose_dict.keys())
        missing = set(frames) - set(pose_dict.keys())
        if missing:
            raise ValueError(f"Missing poses for links: {missing}")
        batch = pose_dict[frames[0]].position.shape[0] // num_goalset
        pos = torch.stack([pose_dict[f].position.reshape(batch, num_goalset, 3) for f in frames], dim=1)
        quat = torch.stack([pose_dict[f].quaternion.reshape(batch, num_goalset, 4) for f in frames], dim=1)
        return cls(frames, pos.unsqueeze(1), quat.unsqueeze(1))


@dataclass
class ToolPoseCriteria:
    """how the pose cost of one tool frame is weighted (reference _src/cost/tool_pose_criteria.py:17-200): factors on the
    (x, y, z, roll, pitch, yaw) terms at the last point of a trajectory and at the points before it, convergence tolerances
    (position [m], rotation [rad]) and whether the error is measured in the goal frame"""

    terminal_pose_axes_weight_factor: Optional[Sequence[float]] = None
    non_terminal_pose_axes_weight_factor: Optional[Sequence[float]] = None
    terminal_pose_convergence_tolerance: Optional[Sequence[float]] = None
    non_terminal_pose_convergence_tolerance: Optional[Sequence[float]] = None
    project_distance_to_goal: bool = False

    def __post_init__(self):
        def vec(x, n, default, what):
            x = [default] * n if x is None else [float(v) for v in (x.tolist() if torch.is_tensor(x) else x)]
            if len(x) != n:
                raise ValueError(f"{what} must be a list of {n} floats, got {x}")
            return x

        self.terminal_pose_axes_weight_factor = vec(self.terminal_pose_axes_weight_factor, 6, 1.0, "terminal_pose_axes_weight_factor")
        self.non_terminal_pose_axes_weight_factor = vec(self.non_terminal_pose_axes_weight_factor, 6, 0.0,
                                                        "non_terminal_pose_axes_weight_factor")
        self.terminal_pose_convergence_tolerance = vec(self.terminal_pose_convergence_tolerance, 2, 0.0,
                                                       "terminal_pose_convergence_tolerance")
        self.non_terminal_pose_convergence_tolerance = vec(self.non_terminal_pose_convergence_tolerance, 2, 0.0,
                                                           "non_terminal_pose_convergence_tolerance")
        if torch.is_tensor(self.project_distance_to_goal):
            self.project_distance_to_goal = bool(self.project_distance_to_goal.reshape(-1)[0].item())
        if not isinstance(self.project_distance_to_goal, bool):
            raise ValueError(f"project_distance_to_goal must be a bool, got {self.project_distance_to_goal}")

    def clone(self) -> "ToolPoseCriteria":
        return ToolPoseCriteria(list(self.terminal_pose_axes_weight_factor), list(self.non_terminal_pose_axes_weight_factor),
                                list(self.terminal_pose_convergence_tolerance), list(self.non_terminal_pose_convergence_tolerance),
                                self.project_distance_to_goal)

    @staticmethod
    def track_position(xyz: Sequence[float] = (1.0, 1.0, 1.0)) -> "ToolPoseCriteria":
        w = [xyz[0], xyz[1], xyz[2], 0.0, 0.0, 0.0]
        return ToolPoseCriteria(w, list(w))

    @staticmethod
    def track_orientation(rpy: Sequence[float] = (0.001, 0.001, 0.001), non_terminal_scale: float = 1.0) -> "ToolPoseCriteria":
        return ToolPoseCriteria([0.0, 0.0, 0.0, rpy[0], rpy[1], rpy[2]], [0.0, 0.0, 0.0] + [non_terminal_scale * r for r in rpy])

    @staticmethod
    def track_position_and_orientation(xyz: Sequence[float] = (1.0, 1.0, 1.0), rpy: Sequence[float] = (1.0, 1.0, 1.0),
                                       non_terminal_scale: float = 0.1) -> "ToolPoseCriteria":
        w = [xyz[0], xyz[1], xyz[2], rpy[0], rpy[1], rpy[2]]
        return ToolPoseCriteria(w, [non_terminal_scale * v for v in w])

    @staticmethod
    def linear_motion(axis: str = "z", non_terminal_scale: float = 1.0, project_distance_to_goal: bool = True) -> "ToolPoseCriteria":
        """keep the tool on the line along ``axis`` through the goal (and at the goal's orientation) on the way there: the
        points before the last one are held on the other two axes and on all three rotations"""
        if axis not in ("x", "y", "z"):
            raise ValueError(f"Invalid axis: {axis}, must be 'x', 'y', or 'z'")
        free = [1.0 if axis == a else 0.0 for a in ("x", "y", "z")]
        return ToolPoseCriteria([1.0] * 6, [non_terminal_scale * (1.0 - f) for f in free] + [non_terminal_scale] * 3,
                                project_distance_to_goal=project_distance_to_goal)

    @staticmethod
    def disabled() -> "ToolPoseCriteria":
        """no pose cost on this tool frame (it stays part of the solver's frames): every factor zero (reference :201-215)"""
        return ToolPoseCriteria([0.0] * 6, [0.0] * 6)
