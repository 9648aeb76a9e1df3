"""The rollout hot path: knots -> B-spline -> FK -> self + scene collision -> per-trajectory cost,
and the analytic backward to the knots, as one straight line of kernel launches.

This is the data path of the reference's ``RobotRollout.evaluate_action`` followed by
``cost.backward`` (``curobo/_src/rollout/rollout_robot.py:252-263,537-587`` and
``optim/components/gradient_opt_core.py:445-480``; call stack in SURVEY.md section 3.2) with the
collision cost terms of ``content/configs/task/trajopt/lbfgs_bspline_trajopt.yml:44-53``.  The
reference builds a torch autograd graph (one ``autograd.Function`` per kernel, extra torch kernels
for ``zero_()``, gradient scaling, sphere-gradient add and ``cat_sum``); here forward and VJP are
explicit launches on pre-allocated buffers -- nothing is allocated, synchronised or read back, so
the whole evaluation is hipGraph-capturable, and the two sphere-gradient buffers are summed inside
the FK backward kernel instead of by an elementwise add.

``curobo_amd.hip_ops`` holds the drop-in ``autograd.Function`` wrappers for callers that need
the reference's autograd contract; tests check both paths give the same numbers.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from ..backends import collision as collision_hip
from ..backends import geometry as geometry_hip
from ..backends import kinematics as kinematics_hip
from ..backends import rollout as rollout_hip
from ..backends import trajectory as trajectory_hip
from ..robot.kinematics_params import KinematicsParams
from ..scene.data import SceneData, validate_env_query_idx


@dataclass
class CollisionRolloutCfg:
    """Defaults follow the reference trajopt task (weights/activation: lbfgs_bspline_trajopt.yml
    :44-53; control space BSPLINE_3 with the BASELINE C2 shape: 12 knots x 2 interpolation steps
    -> horizon 32, padded 33)."""

    n_knots: int = 12
    interpolation_steps: int = 2
    bspline_degree: int = 3
    traj_dt: float = 0.05
    self_collision_weight: float = 10000.0
    scene_collision_weight: float = 100000.0
    activation_distance: float = 0.0025
    use_sweep: bool = True
    use_speed_metric: bool = True
    use_self_collision: bool = True
    use_scene_collision: bool = True
    #: one fused launch (csrc/rollout_fused.hip) instead of the 7-kernel sequence whenever one
    #: trajectory fits in LDS; False forces the drop-in kernel sequence (and materialises every
    #: intermediate tensor, which the fused path only does on request)
    use_fused: bool = True
    # fused launches map workgroups to trajectories longest-first (durations measured by the previous
    # launch; same outputs, shorter launch tail when trajectories differ in collision work)
    longest_first_dispatch: bool = True
    #: fused path: also write position[B,H,D] and robot_spheres[B,H,S,4] to HBM
    fused_materialize: bool = False
    #: compile a compile-time shape of the fused launch for THIS robot / horizon at first use when the library holds none
    #: (``backends/fused_jit.py``: hipcc at run time, 5-10 s once, cached on disk -- the reference compiles its kernels per robot
    #: with NVRTC); also switched on by CUROBO_HIP_JIT_SHAPES=1.  Same results, ~20 % faster launches.
    jit_shape: bool = False

    @property
    def horizon(self) -> int:
        return (self.n_knots + self.bspline_degree + 1) * self.interpolation_steps

    @property
    def padded_horizon(self) -> int:
        return self.horizon + 1


class CollisionRollout:
    """Cost and gradient of ``batch_size`` B-spline trajectories (one "rollout" each)."""

    def __init__(self, kin: KinematicsParams, scene: Optional[SceneData], batch_size: int,
                 cfg: Optional[CollisionRolloutCfg] = None):
        self.kin = kin
        self.scene = scene
        self.cfg = cfg or CollisionRolloutCfg()
        self.device = kin.device
        self.action_horizon = self.cfg.n_knots
        self.action_dim = kin.num_dof
        self.batch_size = 0
        self.use_multi_env = False
        self._fused_ok: Optional[bool] = None
        self._dispatch = None
        d = self.device
        self._w_self = torch.tensor([self.cfg.self_collision_weight], device=d)
        self._w_scene = torch.tensor([self.cfg.scene_collision_weight], device=d)
        self._eta = torch.tensor([self.cfg.activation_distance], device=d)
        self._speed_dt = torch.tensor([self.cfg.traj_dt], device=d)
        self._traj_dt = torch.tensor([self.cfg.traj_dt], device=d)
        self._implicit_goal = torch.zeros(1, dtype=torch.uint8, device=d)
        self.update_batch_size(batch_size)
        self.update_start_state(None)

    # ------------------------------------------------------------------ buffers
    def update_batch_size(self, batch_size: int) -> None:
        if batch_size == self.batch_size:
            return
        B, H, D = batch_size, self.cfg.padded_horizon, self.action_dim
        S, L, T = self.kin.num_spheres, self.kin.num_links, self.kin.num_pose_links
        d = self.device
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=d, dtype=dt)  # noqa: E731
        self.batch_size = B
        # transition (reference StateFromBSplineKnot buffers, transition/fns_state_transition.py:310-472)
        self.position, self.velocity = z(B, H, D), z(B, H, D)
        self.acceleration, self.jerk = z(B, H, D), z(B, H, D)
        self.out_dt = z(B)
        self.start_idx = z(B, dt=torch.int32)
        self.goal_idx = z(B, dt=torch.int32)
        # kinematics (reference KinematicsFusedFunction.create_buffers, cuda_ops/kinematics.py:27-90)
        self.link_pos, self.link_quat = z(B, H, T, 3), z(B, H, T, 4)
        self.robot_spheres = z(B, H, S, 4)
        self.cumul_mat = z(B, H, L, 3, 4)
        self.com = z(B, H, 4)
        self.env_query_idx = z(B, dt=torch.int32)
        # self collision (reference SelfCollisionCost.setup_batch_tensors, cost/cost_self_collision.py:31-89)
        self.self_dist = z(B, H, 1)
        self.self_grad = z(B, H, S, 4)
        self.self_sparse = z(B, H, S, dt=torch.uint8)
        self._pair_distance = z(1)
        self._bbmv = z(1)
        self._bbmi = z(2, dt=torch.int16)
        # scene collision (reference CollisionBuffer, geom/collision/buffer_collision.py:25-105)
        self.scene_dist = z(B, H, S)
        self.scene_grad = z(B, H, S, 4)
        # outputs
        self.cost = z(B)
        self.grad_q = z(B, H, D)
        self.grad_zero_pos, self.grad_zero_quat = z(B, H, T, 3), z(B, H, T, 4)
        self.grad_zero_state = z(B, H, D)
        self.grad_knots = z(B, self.cfg.n_knots, D)

    def update_env_query_idx(self, env_query_idx: Optional[torch.Tensor]) -> None:
        """Scene environment of every trajectory (reference ``idxs_env`` / ``use_multi_env`` of the
        collision costs, cost/cost_scene_collision.py:58-198): trajectory b collides against the
        obstacles of environment ``env_query_idx[b]``; ``None`` = every trajectory uses env 0."""
        if env_query_idx is None:
            self.use_multi_env = False
            self.env_query_idx.zero_()
            return
        self.use_multi_env = True
        validate_env_query_idx(env_query_idx, self.scene, self.kin.num_envs)
        self.env_query_idx.copy_(env_query_idx.to(device=self.device, dtype=torch.int32).reshape(-1))

    def update_start_state(self, start_position: Optional[torch.Tensor]) -> None:
        """One shared start state (position; zero velocity/acceleration/jerk)."""
        D, d = self.action_dim, self.device
        if start_position is None:
            start_position = torch.zeros(1, D, device=d)
        self.start_pos = start_position.reshape(-1, D).contiguous().clone()
        n = self.start_pos.shape[0]
        self.start_vel = torch.zeros(n, D, device=d)
        self.start_acc = torch.zeros(n, D, device=d)
        self.start_jerk = torch.zeros(n, D, device=d)
        self.goal_pos = torch.zeros(1, D, device=d)
        self.goal_vel = torch.zeros(1, D, device=d)
        self.goal_acc = torch.zeros(1, D, device=d)
        self.goal_jerk = torch.zeros(1, D, device=d)

    # ------------------------------------------------------------------ forward
    def compute_state_from_action(self, act_seq: torch.Tensor) -> torch.Tensor:
        cfg, B = self.cfg, self.batch_size
        trajectory_hip.launch_bspline_interpolation_forward_kernel(
            self.position, self.velocity, self.acceleration, self.jerk, self.out_dt, act_seq,
            self.start_pos, self.start_vel, self.start_acc, self.start_jerk, self.goal_pos,
            self.goal_vel, self.goal_acc, self.goal_jerk, self.start_idx, self.goal_idx, self._traj_dt,
            self._implicit_goal, B, cfg.padded_horizon, self.action_dim, cfg.n_knots, cfg.bspline_degree)
        return self.position

    def compute_kinematics(self, q: torch.Tensor) -> torch.Tensor:
        k, B, H = self.kin, self.batch_size, self.cfg.padded_horizon
        kinematics_hip.launch_kinematics_forward_spheres(
            self.link_pos, self.link_quat, self.robot_spheres, self.com, self.cumul_mat, q,
            k.fixed_transforms, k.link_spheres, k.link_masses_com, k.joint_map_type, k.joint_map,
            k.link_map, k.tool_frame_map, k.link_sphere_idx_map, k.joint_offset_map, self.env_query_idx,
            k.num_envs, B * H, H, self.action_dim, k.num_spheres, 32, True, False)
        return self.robot_spheres

    def compute_costs(self) -> torch.Tensor:
        cfg, k, B, H = self.cfg, self.kin, self.batch_size, self.cfg.padded_horizon
        S = k.num_spheres
        if cfg.use_self_collision:
            sc = k.self_collision
            geometry_hip.self_collision_distance(
                self.self_dist, self.self_grad, self._pair_distance, self.self_sparse, self.robot_spheres,
                sc.sphere_padding, self._w_self, sc.collision_pairs, self._bbmv, self._bbmi,
                sc.num_blocks_per_batch, sc.max_threads_per_block, B, H, S, sc.collision_pairs.shape[0],
                False, True)
        if cfg.use_scene_collision and self.scene is not None:
            collision_hip.sphere_obstacle_collision(
                self.scene_dist, self.scene_grad, self.robot_spheres, self.scene.struct, self._w_scene,
                self._eta, self.env_query_idx, B, H, S, self.use_multi_env, 3 if cfg.use_sweep else 0,
                cfg.use_sweep and cfg.use_speed_metric, self._speed_dt)
        collision_hip.trajectory_cost_sum(
            self.cost, self.self_dist if cfg.use_self_collision else None,
            self.scene_dist if (cfg.use_scene_collision and self.scene is not None) else None, B, H, S)
        return self.cost

    def evaluate_action(self, act_seq: torch.Tensor) -> torch.Tensor:
        """cost[B] of ``act_seq[B, n_knots, D]`` (reference RobotRollout.evaluate_action)."""
        self.compute_kinematics(self.compute_state_from_action(act_seq))
        return self.compute_costs()

    # ------------------------------------------------------------------ reference Rollout protocol
    # (curobo/_src/rollout/rollout_protocol.py:46-174): the members solvers and optimisers rely on
    @property
    def action_bound_lows(self) -> torch.Tensor:
        return self.kin.joint_limits_position[0]

    @property
    def action_bound_highs(self) -> torch.Tensor:
        return self.kin.joint_limits_position[1]

    @property
    def dt(self) -> float:
        return self.cfg.traj_dt

    @property
    def sum_horizon(self) -> bool:
        return True  # costs are returned summed over the horizon, one value per trajectory

    def compute_metrics_from_action(self, act_seq: torch.Tensor) -> dict:
        """reference :107-121: evaluate through the kernel sequence (materialised state) and report
        per-trajectory metrics: cost, worst self / scene collision terms, feasibility."""
        with torch.no_grad():
            cost = self.evaluate_action(act_seq.view(self.batch_size, self.cfg.n_knots, self.action_dim)).clone()
        B = self.batch_size
        self_c = self.self_dist.view(B, -1).sum(-1) if self.cfg.use_self_collision else torch.zeros_like(cost)
        scene_on = self.cfg.use_scene_collision and self.scene is not None
        scene_c = self.scene_dist.view(B, -1).sum(-1) if scene_on else torch.zeros_like(cost)
        return {"cost": cost, "self_collision_cost": self_c.clone(), "scene_collision_cost": scene_c.clone(),
                "feasible": (self_c + scene_c) == 0.0, "position": self.position}

    def update_params(self, start_position: Optional[torch.Tensor] = None, scene: Optional[SceneData] = None,
                      env_query_idx: Optional[torch.Tensor] = None) -> bool:
        """reference :123-129 (goal / start / world updates between solves, buffers stay in place)"""
        if start_position is not None:
            self.update_start_state(start_position)
        if scene is not None:
            self.scene = scene
            self._fused_ok = None
        if env_query_idx is not None:
            self.update_env_query_idx(env_query_idx)
        return True

    def reset(self, **kwargs) -> bool:
        return True

    def reset_shape(self) -> bool:
        return True

    def reset_seed(self) -> None:
        return None

    # ------------------------------------------------------------------ backward
    def backward(self) -> torch.Tensor:
        """d(sum cost)/d(knots) of the last ``evaluate_action`` (grad_output = 1 per trajectory,
        the reference's ``cost.backward(gradient=self._l_vec)`` with ``_l_vec`` = ones)."""
        cfg, k, B, H = self.cfg, self.kin, self.batch_size, self.cfg.padded_horizon
        use_scene = cfg.use_scene_collision and self.scene is not None
        ga = self.self_grad if cfg.use_self_collision else (self.scene_grad if use_scene else None)
        gb = self.scene_grad if (cfg.use_self_collision and use_scene) else None
        kinematics_hip.launch_kinematics_backward(
            self.grad_q, self.grad_zero_pos, self.grad_zero_quat, ga, self.com, self.com, self.grad_zero_pos,
            self.cumul_mat, k.link_spheres, k.link_masses_com, k.link_map, k.joint_map, k.joint_map_type,
            k.tool_frame_map, k.link_sphere_idx_map, k.link_chain_data, k.link_chain_offsets,
            k.joint_links_data, k.joint_links_offsets, k.joint_affects_endeffector, k.joint_offset_map,
            self.env_query_idx, k.num_envs, B * H, H, self.action_dim, k.num_spheres if ga is not None else 0,
            False, False, grad_spheres_b=gb)
        trajectory_hip.launch_bspline_interpolation_backward_kernel(
            self.grad_knots, self.grad_q, self.grad_zero_state, self.grad_zero_state, self.grad_zero_state,
            self._traj_dt, self.goal_idx, self._implicit_goal, B, H, self.action_dim, cfg.n_knots,
            cfg.bspline_degree, False)
        return self.grad_knots

    # ------------------------------------------------------------------ fused
    def fused_available(self) -> bool:
        cfg, k = self.cfg, self.kin
        use_scene = cfg.use_scene_collision and self.scene is not None
        n_obs = (self.scene.struct.max_cuboids + self.scene.struct.max_voxel_grids) if use_scene else 0
        n_pairs = k.self_collision.collision_pairs.shape[0] if cfg.use_self_collision else 0
        need = rollout_hip.rollout_trajectory_fused_lds_bytes(
            cfg.padded_horizon, self.action_dim, k.num_links, k.num_spheres, n_pairs,
            int(k.link_chain_data.shape[0]), n_obs)
        if use_scene and getattr(self.scene.struct, "mesh_set", None) is not None:
            return False  # mesh obstacles are queried by their own launch (BVH): the kernel sequence runs
        return need <= rollout_hip.FUSED_LDS_LIMIT and k.num_links <= 128

    def _maybe_jit_shape(self) -> None:
        """cfg.jit_shape / CUROBO_HIP_JIT_SHAPES: a compile-time shape for this rollout's dimensions, built once when the library
        has none (never inside a captured launch sequence: the first call of a rollout is an eager warm-up)"""
        from ..backends import fused_jit

        if not (self.cfg.jit_shape or fused_jit.enabled_by_env()) or not self.cfg.use_self_collision:
            return
        k, cfg = self.kin, self.cfg
        lanes = getattr(k.self_collision.collision_pairs, "_self_lane_lists", None)
        use_scene = cfg.use_scene_collision and self.scene is not None
        n_obs = (self.scene.struct.max_cuboids + self.scene.struct.max_voxel_grids) if use_scene else 0
        fused_jit.ensure_shape(cfg.padded_horizon, cfg.n_knots, self.action_dim, k.num_links, k.num_spheres,
                               int(k.self_collision.collision_pairs.shape[0]), int(k.link_chain_data.shape[0]),
                               int(lanes[1]) if lanes is not None else 0, n_obs, with_trajopt_terms=False)

    def _dispatch_order(self):
        """longest-first dispatch workspace of this rollout's fused launches (cfg.longest_first_dispatch)"""
        if not self.cfg.longest_first_dispatch:
            return None
        if self._dispatch is None:
            self._dispatch = rollout_hip.DispatchOrder(self.batch_size, self.cost.device)
        return self._dispatch

    def cost_and_gradient_fused(self, act_seq: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Same numbers as ``evaluate_action`` + ``backward`` from one kernel launch."""
        cfg, k, B = self.cfg, self.kin, self.batch_size
        use_scene = cfg.use_scene_collision and self.scene is not None
        sc = k.self_collision
        mat = cfg.fused_materialize
        rollout_hip.rollout_trajectory_fused(
            self.cost, self.grad_knots, self.position if mat else None, self.robot_spheres if mat else None,
            act_seq, self.start_pos, self.start_vel, self.start_acc, self.start_jerk, self.goal_pos,
            self.goal_vel, self.goal_acc, self.goal_jerk, self.start_idx, self.goal_idx, self._traj_dt,
            self._implicit_goal, k.fixed_transforms, k.link_spheres, k.joint_map_type, k.joint_map, k.link_map,
            k.link_sphere_idx_map, k.link_chain_data, k.link_chain_offsets, k.joint_offset_map,
            sc.sphere_padding, self._w_self if cfg.use_self_collision else None,
            sc.collision_pairs if cfg.use_self_collision else None,
            self.scene.struct if use_scene else None, self._w_scene if use_scene else None,
            self._eta, self._speed_dt, self.env_query_idx, k.num_envs, self.use_multi_env, B, cfg.padded_horizon,
            self.action_dim, cfg.n_knots, cfg.bspline_degree, 3 if cfg.use_sweep else 0,
            cfg.use_sweep and cfg.use_speed_metric, self._dispatch_order())
        return self.cost, self.grad_knots

    def cost_and_gradient(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x[B, n_knots*D] -> (cost[B], grad[B, n_knots*D]); buffers are reused every call."""
        act = x.view(self.batch_size, self.cfg.n_knots, self.action_dim)
        if self.cfg.use_fused:
            if self._fused_ok is None:
                self._fused_ok = self.fused_available()
                if self._fused_ok:
                    self._maybe_jit_shape()
            if self._fused_ok:
                cost, grad = self.cost_and_gradient_fused(act)
                return cost, grad.view(self.batch_size, -1)
        cost = self.evaluate_action(act)
        grad = self.backward()
        return cost, grad.view(self.batch_size, -1)

    # ------------------------------------------------------------------ accounting
    def algorithmic_bytes_per_point(self) -> int:
        """8D + 56T + 96L + 84S + 4 (API-materialised tensors, fp32)."""
        k = self.kin
        return 8 * k.num_dof + 56 * k.num_pose_links + 96 * k.num_links + 84 * k.num_spheres + 4
