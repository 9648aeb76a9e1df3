"""Teleport (horizon 1) rollout for inverse kinematics: q -> FK -> tool-pose + c-space bound +
self + scene collision costs, and the analytic gradient back to q.

Data path of the reference's ``RobotRollout`` with ``StateFromPositionTeleport``
(``content/configs/task/ik/transition_ik.yml``) and the IK cost set
(``content/configs/task/ik/lbfgs_ik.yml:3-36``); kernels cited in the backend modules.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch

from ..backends import collision as collision_hip
from ..backends import cost as cost_hip
from ..backends import geometry as geometry_hip
from ..backends import kinematics as kinematics_hip
from ..backends import rollout as rollout_hip
from ..robot.kinematics_params import KinematicsParams
from ..scene.data import SceneData, validate_env_query_idx


@dataclass
class IKRolloutCfg:
    """Defaults = reference ``lbfgs_ik.yml``."""

    pose_weight: List[float] = field(default_factory=lambda: [10000.0, 500.0])
    pose_convergence_tolerance: List[float] = field(default_factory=lambda: [1e-8, 1e-8])
    rotation_method: int = 0  # use_lie_group: false
    cspace_weight: List[float] = field(default_factory=lambda: [5000.0, 0.0])
    cspace_activation_distance: List[float] = field(default_factory=lambda: [0.01, 0.01])
    scene_collision_weight: float = 5000.0
    scene_activation_distance: float = 0.0
    self_collision_weight: float = 5000.0
    #: one fused launch (csrc/rollout_fused.hip, rollout_ik_fused_kernel) for cost + gradient when 16
    #: configurations fit in LDS; False = the seven drop-in launches
    use_fused: bool = True


class IKRollout:
    """cost[B] and d cost / d q [B, D] for B joint configurations against per-row goal poses."""

    def __init__(self, kin: KinematicsParams, scene: Optional[SceneData], batch_size: int,
                 cfg: Optional[IKRolloutCfg] = None, num_goalset: int = 1):
        self.kin, self.scene, self.cfg = kin, scene, cfg or IKRolloutCfg()
        self.device = kin.device
        self.action_horizon, self.action_dim = 1, kin.num_dof
        self.num_goalset = num_goalset
        d, T, D = self.device, kin.num_pose_links, kin.num_dof
        c = self.cfg
        f = lambda v: torch.tensor(v, device=d, dtype=torch.float32)  # noqa: E731
        self._pose_w = f(c.pose_weight)
        self._axes_w = torch.ones(T, 6, device=d)
        self._tol = f([c.pose_convergence_tolerance] * T)
        self._project = torch.zeros(T, dtype=torch.uint8, device=d)
        self._cs_w, self._cs_eta = f(c.cspace_weight), f(c.cspace_activation_distance)
        self._p_b = kin.joint_limits_position.contiguous()
        self._effort_b = torch.stack([torch.full((D,), -1e9, device=d), torch.full((D,), 1e9, device=d)])
        self._v_b = kin.joint_limits_velocity.contiguous()
        self._zero1 = torch.zeros(1, device=d)
        self._zeroD = torch.zeros(1, D, device=d)
        self._onesD = torch.ones(D, device=d)
        self._reg = torch.zeros(2, device=d)
        self._w_scene, self._eta_scene = f([c.scene_collision_weight]), f([c.scene_activation_distance])
        self._w_self = f([c.self_collision_weight])
        self.batch_size = 0
        self._fused_ok: Optional[bool] = None
        self._env_runs_ok = True  # env_query_idx constant over aligned runs of 16 rows (update_env_query_idx)
        self.update_batch_size(batch_size)

    def update_batch_size(self, B: int) -> None:
        if B == self.batch_size:
            return
        k, d = self.kin, self.device
        T, S, L, D = k.num_pose_links, k.num_spheres, k.num_links, k.num_dof
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=d, dtype=dt)  # noqa: E731
        self.batch_size = B
        self.link_pos, self.link_quat = z(B, 1, T, 3), z(B, 1, T, 4)
        self.robot_spheres, self.cumul_mat, self.com = z(B, 1, S, 4), z(B, 1, L, 3, 4), z(B, 1, 4)
        self.env_query_idx = z(B, dt=torch.int32)
        self.pose_cost, self.pose_pos_dist, self.pose_rot_dist = z(B, 1, 2 * T), z(B, 1, T), z(B, 1, T)
        self.pose_grad_pos, self.pose_grad_quat = z(B, 1, T, 3), z(B, 1, T, 4)
        self.goalset_idx = z(B, 1, T, dt=torch.int32)
        self.cspace_cost, self.cspace_grad = z(B, 1, D), z(B, 1, D)
        self.self_dist, self.self_grad = z(B, 1, 1), z(B, 1, S, 4)
        self.self_sparse = z(B, 1, S, dt=torch.uint8)
        self.scene_dist, self.scene_grad = z(B, 1, S), z(B, 1, S, 4)
        self._pd, self._bbmv, self._bbmi = z(1), z(1), z(2, dt=torch.int16)
        self.cost, self.grad_q = z(B), z(B, 1, D)
        self.idxs_goal = z(B, dt=torch.int32)
        self._idxs_src = None  # (the row -> goal map was reallocated: the next update_goals must copy, whatever tensor it is handed)
        self._idx0 = z(B, dt=torch.int32)
        self.goal_position = z(1, T, self.num_goalset, 3)
        self.goal_quat = z(1, T, self.num_goalset, 4)
        self.goal_quat[..., 0] = 1.0

    use_multi_env = False

    def update_env_query_idx(self, env_query_idx: Optional[torch.Tensor]) -> None:
        """Scene environment of every configuration (reference ``idxs_env`` / ``use_multi_env``,
        cost/cost_scene_collision.py:58-198): row b collides with the obstacles of environment
        ``env_query_idx[b]``; ``None`` = env 0.  The fused IK launch serves 16 configurations per workgroup from one
        staged scene / sphere set: it runs when the index is constant over aligned runs of 16 rows (the seeds of one
        problem; checked here, one host read-back, never inside a captured launch sequence), else the kernel
        sequence does.  Switching modes changes the launches: re-capture graphs."""
        self.use_multi_env = env_query_idx is not None
        if env_query_idx is None:
            self.env_query_idx.zero_()
            self._env_runs_ok = True
        else:
            validate_env_query_idx(env_query_idx, self.scene, self.kin.num_envs)
            self.env_query_idx.copy_(env_query_idx.to(device=self.device, dtype=torch.int32).reshape(-1))
            idx = self.env_query_idx
            n = idx.numel()
            first = idx[(torch.arange(n, device=idx.device) // 16) * 16]
            self._env_runs_ok = bool((idx == first).all())

    def update_goals(self, goal_position: torch.Tensor, goal_quat: torch.Tensor, idxs_goal: torch.Tensor) -> None:
        """goal_position [G, T, num_goalset, 3], goal_quat (wxyz) [G, T, num_goalset, 4], idxs_goal [B]."""
        assert goal_position.shape[1:3] == (self.kin.num_pose_links, self.num_goalset)
        if goal_position.shape == self.goal_position.shape:
            self.goal_position.copy_(goal_position)
            self.goal_quat.copy_(goal_quat)
        else:  # re-allocation invalidates captured graphs: callers re-capture after a shape change
            self.goal_position = goal_position.to(self.device, torch.float32).contiguous().clone()
            self.goal_quat = goal_quat.to(self.device, torch.float32).contiguous().clone()
        key = (idxs_goal, idxs_goal._version)  # (solvers pass the same, unmodified row -> goal map every solve: no copy then)
        last = getattr(self, "_idxs_src", None)
        if last is None or last[0] is not key[0] or last[1] != key[1]:
            self.idxs_goal.copy_(idxs_goal.to(torch.int32))
            self._idxs_src = key

    def update_tool_pose_criteria(self, criteria) -> None:
        """``{tool frame: ToolPoseCriteria}``: an IK rollout has one point per row, so only the terminal factors, the
        terminal tolerance and the projection flag apply (reference ToolPoseCost.update_tool_pose_criteria); in place"""
        for name, c in criteria.items():
            if name not in self.kin.tool_frames:
                raise ValueError(f"tool frame {name} not in {self.kin.tool_frames}")
            i, f = self.kin.tool_frames.index(name), lambda v: torch.tensor(v, device=self.device, dtype=torch.float32)  # noqa: E731
            self._axes_w[i].copy_(f(c.terminal_pose_axes_weight_factor))
            self._tol[i].copy_(f(c.terminal_pose_convergence_tolerance))
            self._project[i] = int(bool(c.project_distance_to_goal))

    # ------------------------------------------------------------------ forward + backward
    def evaluate(self, q: torch.Tensor, with_gradient: bool = True) -> torch.Tensor:
        k, B, c = self.kin, self.batch_size, self.cfg
        T, S, D = k.num_pose_links, k.num_spheres, k.num_dof
        kinematics_hip.launch_kinematics_forward_spheres(
            self.link_pos, self.link_quat, self.robot_spheres, self.com, self.cumul_mat, q, k.fixed_transforms,
            k.link_spheres, k.link_masses_com, k.joint_map_type, k.joint_map, k.link_map, k.tool_frame_map,
            k.link_sphere_idx_map, k.joint_offset_map, self.env_query_idx, k.num_envs, B, 1, D, S, 32, True, False)
        cost_hip.tool_pose_distance(
            self.pose_cost, self.pose_pos_dist, self.pose_rot_dist, self.pose_grad_pos, self.pose_grad_quat,
            self.goalset_idx, self.link_pos, self.link_quat, self.goal_position, self.goal_quat, self.idxs_goal,
            self._pose_w, self._axes_w, self._axes_w, self._tol, self._tol, self._project, B, 1, T,
            self.num_goalset, c.rotation_method)
        cost_hip.cspace_position_cost(
            self.cspace_cost, self.cspace_grad, None, q, None, self._zeroD, self._idx0, self._p_b, self._effort_b,
            self._cs_w, self._cs_eta, self._zero1, self._onesD, self._reg, self._zeroD, self._zeroD, self._idx0,
            self._v_b, self._zero1, True, B, 1, D)
        sc = k.self_collision
        geometry_hip.self_collision_distance(
            self.self_dist, self.self_grad, self._pd, self.self_sparse, self.robot_spheres, sc.sphere_padding,
            self._w_self, sc.collision_pairs, self._bbmv, self._bbmi, 1, 256, B, 1, S, sc.collision_pairs.shape[0],
            False, True)
        use_scene = self.scene is not None
        if use_scene:
            collision_hip.sphere_obstacle_collision(
                self.scene_dist, self.scene_grad, self.robot_spheres, self.scene.struct, self._w_scene,
                self._eta_scene, self.env_query_idx, B, 1, S, self.use_multi_env, 0, False, None)
        if with_gradient:
            kinematics_hip.launch_kinematics_backward(
                self.grad_q, self.pose_grad_pos, self.pose_grad_quat, self.self_grad, self.com, self.com,
                self.pose_grad_pos, self.cumul_mat, k.link_spheres, k.link_masses_com, k.link_map, k.joint_map,
                k.joint_map_type, k.tool_frame_map, k.link_sphere_idx_map, k.link_chain_data, k.link_chain_offsets,
                k.joint_links_data, k.joint_links_offsets, k.joint_affects_endeffector, k.joint_offset_map,
                self.env_query_idx, k.num_envs, B, 1, D, S, False, False,
                grad_spheres_b=self.scene_grad if use_scene else None)
        cost_hip.rollout_point_aggregate(
            self.cost, self.grad_q if with_gradient else None, self.pose_cost, self.cspace_cost,
            self.cspace_grad if with_gradient else None, self.self_dist, self.scene_dist if use_scene else None, B, T,
            D, S)
        return self.cost

    # ------------------------------------------------------------------ fused
    def fused_available(self) -> bool:
        k = self.kin
        n_obs = (self.scene.struct.max_cuboids + self.scene.struct.max_voxel_grids) if self.scene is not None else 0
        need = rollout_hip.rollout_ik_fused_lds_bytes(
            k.num_dof, k.num_links, k.num_spheres, int(k.self_collision.collision_pairs.shape[0]),
            int(k.link_chain_data.shape[0]), n_obs)
        if self.scene is not None and getattr(self.scene.struct, "mesh_set", None) is not None:
            return False  # mesh obstacles are queried by their own launch (BVH): the kernel sequence runs
        return need <= rollout_hip.FUSED_LDS_LIMIT and k.num_links <= 128 and k.num_dof <= 64

    def cost_and_gradient_fused(self, q: torch.Tensor, with_metrics: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """Same numbers as ``evaluate`` from one launch; ``with_metrics`` also fills the pose-error,
        link-pose and sphere buffers the solver's metrics read."""
        k, B, c = self.kin, self.batch_size, self.cfg
        sc = k.self_collision
        m = with_metrics
        rollout_hip.rollout_ik_fused(
            self.cost, self.grad_q, self.pose_cost if m else None, self.pose_pos_dist if m else None,
            self.pose_rot_dist if m else None, self.goalset_idx if m else None, self.link_pos if m else None,
            self.link_quat if m else None, self.robot_spheres if m else None, self.cspace_cost if m else None,
            q, self.goal_position, self.goal_quat, self.idxs_goal, self._pose_w, self._axes_w, self._tol, self._project,
            self.num_goalset, c.rotation_method, self._p_b, self._cs_w, self._cs_eta, k.fixed_transforms, k.link_spheres,
            k.joint_map_type, k.joint_map, k.link_map, k.tool_frame_map, k.link_sphere_idx_map, k.link_chain_data,
            k.link_chain_offsets, k.joint_offset_map, sc.sphere_padding, self._w_self, sc.collision_pairs,
            self.scene.struct if self.scene is not None else None, self._w_scene, self._eta_scene, B, k.num_dof,
            self.env_query_idx, k.num_envs, self.use_multi_env)
        return self.cost, self.grad_q.view(B, -1)

    def cost_and_gradient(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x[B, D] -> (cost[B], grad[B, D]) in static buffers (graph friendly)."""
        if self.cfg.use_fused and (self._env_runs_ok or not (self.use_multi_env or self.kin.num_envs > 1)):
            if self._fused_ok is None:
                self._fused_ok = self.fused_available()
            if self._fused_ok:
                return self.cost_and_gradient_fused(x.view(self.batch_size, self.action_dim).contiguous())
        cost = self.evaluate(x.view(self.batch_size, 1, self.action_dim), with_gradient=True)
        return cost, self.grad_q.view(self.batch_size, -1)
