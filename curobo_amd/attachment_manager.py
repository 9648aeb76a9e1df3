"""Attaching / detaching obstacles to a robot link: a grasped object becomes spheres of the ``attached_object`` link, which the
collision kernels then carry through every rollout like the robot's own.

Counterpart of the reference's ``AttachmentManager`` (``curobo/_src/collision/attachment_manager.py:24-340``): the same
members with the same meaning -- ``fit_spheres`` (once), ``update`` (per-environment obstacle-to-link offsets from one batched
FK call, written into ``KinematicsParams.link_spheres[env, slots]`` in place, unused slots disabled with radius -100),
``attach`` / ``attach_from_scene`` (+ switching the world's copy of the object off), ``detach`` (the link's spheres as loaded,
the world obstacles back on).  Every write lands in tensors the kernels and captured graphs already hold.

Sphere fitting is NOT the reference's: its ``fit_spheres_to_mesh`` (MorphIt / voxel fits over a ``trimesh`` mesh,
``geom/sphere_fit/``) is an offline geometry tool outside the hot path and needs ``trimesh``; here cuboids, spheres, capsules
and cylinders get a closed-form lattice of inscribed spheres (``fit_spheres_to_obstacle``; ``conservative=True``: of COVERING spheres),
meshes and voxel grids the lattice of their bounding cuboid.  A caller with its own fit hands the [n, 4] tensor to ``update`` directly, as with the reference.
"""

from __future__ import annotations

import math
import time
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from .types import DeviceCfg, JointState, Pose


@dataclass
class SphereFitResult:
    """what the reference's fit hands back (geom/sphere_fit/types.py): centres [n, 3], radii [n], count, wall time"""

    centers: torch.Tensor
    radii: torch.Tensor
    num_spheres: int
    fit_time_s: float


def _lattice(half: np.ndarray, radius: float, budget: Optional[int]) -> np.ndarray:
    """centres of spheres of ``radius`` inside the box [-half, half]: per axis from -half + r to half - r; the pitch is the
    smallest of r .. that keeps the count within ``budget`` (pitch <= 2 r leaves no gap along an axis)"""
    span = np.maximum(half - radius, 0.0)

    def counts(pitch):
        return [1 if s < 1e-9 else int(math.ceil(2.0 * s / pitch - 1e-9)) + 1 for s in span]

    pitch = radius
    if budget is not None:
        budget = max(int(budget), 1)
        while int(np.prod(counts(pitch))) > budget:
            pitch *= 1.1
    axes = [np.zeros(1) if n == 1 else np.linspace(-s, s, n) for s, n in zip(span, counts(pitch))]
    return np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, 3)


def _covering_lattice(half: np.ndarray, budget: Optional[int]):
    """(centres, radius) of equal spheres whose UNION contains the box [-half, half]: the box is cut into nx x ny x nz equal cells
    (as cubic as the budget allows; default: cells of about the smallest extent), one sphere per cell centre with the cell's half
    diagonal as radius"""
    ext = 2.0 * np.maximum(half, 1e-9)
    budget = max(int(budget), 1) if budget is not None else 64
    pitch = float(ext.min())
    counts = lambda p: np.maximum(np.ceil(ext / p - 1e-9).astype(int), 1)  # noqa: E731
    while int(np.prod(counts(pitch))) > budget:
        pitch *= 1.1
    n = counts(pitch)
    cell = ext / n
    axes = [(-half[a] + cell[a] * (np.arange(n[a]) + 0.5)) for a in range(3)]
    return np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, 3), 0.5 * float(np.linalg.norm(cell))


def fit_spheres_to_obstacle(obstacle, num_spheres: Optional[int] = None, surface_radius: float = 0.002, conservative: bool = False) -> np.ndarray:
    """[n, 4] spheres (x y z r) in the WORLD frame (the obstacle's pose applied, as the reference's combined mesh is built with
    ``transform_with_pose=True``) that lie inside the obstacle and touch its faces: a sphere is itself; a capsule a row of spheres of
    its radius; everything else a lattice of spheres of radius = the smallest half extent of its (bounding) cuboid, never less than
    ``surface_radius``.  ``conservative``: the spheres COVER the obstacle's (bounding) cuboid instead of lying inside it (they stick out by
    less than their radius): for a grasped object that must not touch anything."""
    from .scene.types import Capsule, Pose7, Sphere

    if isinstance(obstacle, Sphere):
        local = np.array([[0.0, 0.0, 0.0, float(obstacle.radius)]])
    elif isinstance(obstacle, Capsule):
        base, tip = np.asarray(obstacle.base, np.float64), np.asarray(obstacle.tip, np.float64)
        r = float(obstacle.radius)
        n = max(int(math.ceil(np.linalg.norm(tip - base) / r)) + 1, 1)
        if num_spheres is not None:
            n = max(min(n, int(num_spheres)), 1)
        pts = base[None] + (tip - base)[None] * (np.linspace(0.0, 1.0, n)[:, None] if n > 1 else np.full((1, 1), 0.5))
        local = np.concatenate([pts, np.full((n, 1), r)], axis=1)
    else:
        half = 0.5 * np.asarray(obstacle.get_cuboid().dims, np.float64)
        if conservative:
            c, r = _covering_lattice(half, num_spheres)
        else:
            r = max(float(half.min()), float(surface_radius))
            c = _lattice(half, r, num_spheres)
        local = np.concatenate([c, np.full((c.shape[0], 1), r)], axis=1)
    pose = obstacle.pose if obstacle.pose is not None else [0, 0, 0, 1, 0, 0, 0]
    if not isinstance(obstacle, (Sphere, Capsule)):
        pose = obstacle.get_cuboid().pose  # (a cylinder's / mesh's bounding cuboid carries its own pose)
    out = local.copy()
    out[:, :3] = Pose7(pose).transform(local[:, :3])
    return out.astype(np.float32)


class AttachmentManager:
    """see the module docstring; ``kinematics`` = ``curobo_amd.kinematics.Kinematics``, ``scene_collision`` = the ``SceneData``
    of the world (or anything with ``enable_obstacle(name, enable, env_idx)`` and ``scene_model``) or ``None``"""

    def __init__(self, kinematics, scene_collision=None, device_cfg: Optional[DeviceCfg] = None):
        self._kinematics = kinematics
        self._scene_collision = scene_collision
        self._device_cfg = device_cfg or DeviceCfg(device=kinematics.kinematics_config.link_spheres.device)
        self._last_fit_result: Optional[SphereFitResult] = None
        self._attached_link_name: Optional[str] = None
        self._disabled_obstacle_names: List[str] = []
        self._disabled_num_envs: int = 0

    @property
    def kinematics_params(self):
        """the parameters that hold ``link_spheres``"""
        return self._kinematics.kinematics_config

    def update_world(self, scene_collision) -> None:
        """a planner whose world was replaced hands the new one over (obstacle switching acts on it from now on)"""
        self._scene_collision = scene_collision

    # ------------------------------------------------------------------ fit
    def fit_spheres(self, obstacles: List, num_spheres: Optional[int] = None, surface_radius: float = 0.002,
                    sphere_fit_type=None, conservative: bool = False) -> torch.Tensor:
        """[n, 4] spheres over all ``obstacles`` (``num_spheres`` shared between them by volume); ``sphere_fit_type`` is accepted
        for the reference's signature and ignored (one closed-form fit here); ``conservative``: covering instead of inscribed spheres"""
        t0 = time.perf_counter()
        if not obstacles:
            raise ValueError("fit_spheres needs at least one obstacle")
        vol = np.array([max(float(np.prod(o.get_cuboid().dims)), 1e-12) for o in obstacles])
        share = [None] * len(obstacles) if num_spheres is None else [max(int(num_spheres * v / vol.sum()), 1) for v in vol]
        sph = np.concatenate([fit_spheres_to_obstacle(o, n, surface_radius, conservative) for o, n in zip(obstacles, share)], axis=0)
        t = self._device_cfg.to_device(sph)
        self._last_fit_result = SphereFitResult(t[:, :3], t[:, 3], int(t.shape[0]), time.perf_counter() - t0)
        return t

    # ------------------------------------------------------------------ write
    def update(self, sphere_tensor: torch.Tensor, joint_states: JointState, link_name: str = "attached_object",
               world_objects_pose_offset: Optional[Pose] = None) -> None:
        """``sphere_tensor`` [n, 4] in the obstacle frame -> the slots of ``link_name`` in every environment.  With
        ``world_objects_pose_offset`` (the obstacle's world pose, [num_envs] or [1]) the spheres are moved into the frame of
        the first tool frame at ``joint_states`` [num_envs, dof] (one batched FK call); without it they are taken as link-local"""
        q = joint_states.position if hasattr(joint_states, "position") else joint_states
        if q.dim() == 1:
            q = q.unsqueeze(0)
        num_envs = int(q.shape[0])
        kp = self.kinematics_params
        slots = kp.get_sphere_index_from_link_name(link_name)
        n_slots, n_fit = int(slots.shape[0]), int(sphere_tensor.shape[0])
        if n_fit > n_slots:
            raise ValueError(f"Fitted {n_fit} spheres but link '{link_name}' only has {n_slots} sphere slots. Reduce num_spheres or "
                             "increase the link's sphere allocation.")
        if num_envs > int(kp.link_spheres.shape[0]):
            raise ValueError(f"{num_envs} joint states for a model with {int(kp.link_spheres.shape[0])} sphere set(s)")
        dev, dt = kp.link_spheres.device, kp.link_spheres.dtype
        sph = sphere_tensor.to(dev, dt)
        centers = sph[:, :3].unsqueeze(0).expand(num_envs, n_fit, 3)
        if world_objects_pose_offset is not None:
            ee = self._kinematics.get_link_poses(q.to(dev, dt), [self._kinematics.tool_frames[0]])
            ee = Pose(ee.position[:, 0], ee.quaternion[:, 0])
            off = Pose(world_objects_pose_offset.position.to(dev, dt).reshape(-1, 3), world_objects_pose_offset.quaternion.to(dev, dt).reshape(-1, 4))
            if off.position.shape[0] not in (1, num_envs):
                raise ValueError(f"world_objects_pose_offset holds {off.position.shape[0]} poses for {num_envs} environments")
            if off.position.shape[0] == 1 and num_envs > 1:
                off = Pose(off.position.expand(num_envs, 3), off.quaternion.expand(num_envs, 4))
            obj_to_link = ee.inverse().multiply(off)
            centers = obj_to_link.batch_transform_points(centers.contiguous())
        rows = torch.zeros(num_envs, n_slots, 4, device=dev, dtype=dt)
        rows[:, :, 3] = -100.0
        rows[:, :n_fit, :3] = centers
        rows[:, :n_fit, 3] = sph[:, 3]
        kp.link_spheres[:num_envs, slots, :] = rows
        self._attached_link_name = link_name

    def attach(self, joint_states: JointState, obstacles: List, link_name: str = "attached_object", num_spheres: Optional[int] = None,
               surface_radius: float = 0.002, sphere_fit_type=None, world_objects_pose_offset: Optional[Pose] = None,
               disable_obstacle_names: Optional[List[str]] = None, conservative: bool = False) -> None:
        """``fit_spheres`` + ``update`` + the named world obstacles off in every environment"""
        sph = self.fit_spheres(obstacles, num_spheres=num_spheres, surface_radius=surface_radius, sphere_fit_type=sphere_fit_type,
                               conservative=conservative)
        self.update(sph, joint_states, link_name, world_objects_pose_offset)
        if disable_obstacle_names and self._scene_collision is not None:
            n = self._get_num_envs(joint_states)
            for name in disable_obstacle_names:
                for e in range(n):
                    self._scene_collision.enable_obstacle(name, enable=False, env_idx=e)
            self._disabled_obstacle_names = list(disable_obstacle_names)
            self._disabled_num_envs = n

    def attach_from_scene(self, joint_states: JointState, obstacle_names: List[str], link_name: str = "attached_object",
                          num_spheres: Optional[int] = None, surface_radius: float = 0.002, sphere_fit_type=None,
                          world_objects_pose_offset: Optional[Pose] = None, conservative: bool = False) -> None:
        """obstacles looked up by name in the world's description (``scene_collision.scene_model``), attached and switched off"""
        if self._scene_collision is None:
            raise ValueError("attach_from_scene requires scene_collision to be set.")
        model = getattr(self._scene_collision, "scene_model", None)
        if model is None:
            raise ValueError("attach_from_scene requires scene_collision.scene_model to be set.")
        if isinstance(model, (list, tuple)):
            model = model[0]
        obstacles = []
        for name in obstacle_names:
            o = model.get_obstacle(name)
            if o is None:
                raise ValueError(f"Obstacle '{name}' not found in scene_collision.scene_model.")
            obstacles.append(o)
        self.attach(joint_states, obstacles, link_name=link_name, num_spheres=num_spheres, surface_radius=surface_radius,
                    sphere_fit_type=sphere_fit_type, world_objects_pose_offset=world_objects_pose_offset, disable_obstacle_names=obstacle_names,
                    conservative=conservative)

    def detach(self, link_name: Optional[str] = None, enable_obstacle_names: Optional[List[str]] = None) -> None:
        """the link's spheres as loaded, the obstacles the last ``attach`` switched off (or the named ones) back on"""
        link_name = link_name if link_name is not None else self._attached_link_name
        if link_name is None:
            return
        self.kinematics_params.reset_link_spheres(link_name)
        names = enable_obstacle_names or self._disabled_obstacle_names
        if names and self._scene_collision is not None:
            for name in names:
                for e in range(self._disabled_num_envs):
                    self._scene_collision.enable_obstacle(name, enable=True, env_idx=e)
        self._attached_link_name = None
        self._disabled_obstacle_names = []
        self._disabled_num_envs = 0

    @staticmethod
    def _get_num_envs(joint_states) -> int:
        q = joint_states.position if hasattr(joint_states, "position") else joint_states
        return 1 if q.dim() == 1 else int(q.shape[0])
