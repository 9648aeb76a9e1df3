"""The full trajectory-optimisation rollout: B-spline knots -> (q, qd, qdd, qddd) -> FK -> tool-pose
goal cost + c-space STATE cost (bounds, smoothness) + self + swept scene collision, and the analytic
gradient back to the knots.

Cost set and weights = the reference trajopt task ``content/configs/task/trajopt/
lbfgs_bspline_trajopt.yml:38-100`` (``RobotRollout`` with ``StateFromBSplineKnot``; call stack in
SURVEY.md section 3.2).  Like ``CollisionRollout`` the data path is a straight line of launches on
static buffers (no autograd graph, hipGraph capturable); the pose / c-space kernels are the ones of
the IK rollout, evaluated over the whole horizon (the pose cost only acts on the last point: the
non-terminal axes weights are zero, as in the reference config).
"""

from __future__ import annotations

import contextlib

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ..backends import collision as collision_hip
from ..backends import cost as cost_hip
from ..backends import dynamics as dynamics_hip
from ..backends import geometry as geometry_hip
from ..backends import kinematics as kinematics_hip
from ..backends import rollout as rollout_hip
from ..backends import trajectory as trajectory_hip
from ..robot.kinematics_params import KinematicsParams
from ..scene.data import SceneData, validate_env_query_idx
from ..util.stream_scope import inside_forked_stream


_limit_vectors: dict = {}


def joint_limit_vector(value, dof: int, device, name: str = "limit") -> torch.Tensor:
    """a scalar (every joint) or a per-joint list [dof] -> fp32 [dof] of positive limits.  Validated on the host and kept per
    (values, device): the solvers ask for it several times per solve, and a host list -> device tensor with a device-side check
    is a blocking copy plus a synchronisation each time (ten per pose-to-pose solve: they were 6 ms of a 12 ms solve spent with
    the host waiting instead of queueing the next stage).  The returned tensor is shared: do not write to it."""
    if isinstance(value, torch.Tensor):
        vals = tuple(float(x) for x in value.detach().reshape(-1).tolist())
    else:
        vals = tuple(float(x) for x in np.asarray(value, dtype=np.float64).reshape(-1))
    if len(vals) == 1:
        vals = vals * dof
    if len(vals) != dof:
        raise ValueError(f"{name}: one value or one per active joint ({dof}) expected, got {len(vals)}")
    if any(not (x > 0.0) for x in vals):
        raise ValueError(f"{name} must be positive, got {list(vals)}")
    key = (vals, str(torch.device(device)))
    v = _limit_vectors.get(key)
    if v is None:
        v = _limit_vectors[key] = torch.tensor(vals, dtype=torch.float32, device=device)
    return v


@dataclass
class TrajOptRolloutCfg:
    n_knots: int = 12
    interpolation_steps: int = 2
    bspline_degree: int = 3
    traj_dt: float = 0.05
    # constraint_cfg
    self_collision_weight: float = 10000.0
    scene_collision_weight: float = 100000.0
    scene_activation_distance: float = 0.0025
    use_sweep: bool = True
    use_speed_metric: bool = True
    # cost_cfg.tool_pose_cfg
    pose_weight: List[float] = field(default_factory=lambda: [1000000.0, 100000.0])
    pose_convergence_tolerance: List[float] = field(default_factory=lambda: [1e-8, 1e-8])
    rotation_method: int = 0
    #: pose-cost weight factor of the NON-terminal points (0 = the pose goal only acts on the last point, as in the
    #: reference trajopt task; > 0 = tracking along the whole horizon, the reference's MPC task
    #: content/configs/task/mpc/: non_terminal_pose_axes_weight_factor)
    non_terminal_pose_factor: float = 0.0
    # cost_cfg.cspace_cfg (cost_type STATE)
    cspace_weight: List[float] = field(default_factory=lambda: [10000.0, 10000.0, 100.0, 50.0, 100.0])
    cspace_activation_distance: List[float] = field(default_factory=lambda: [0.01] * 5)
    cspace_regularization: List[float] = field(default_factory=lambda: [1000.0, 10000.0, 5.0, 0.0, 10000.0])
    retime_weights: bool = True
    retime_regularization_weights: bool = True
    #: one value for every joint or one per active joint [dof] (reference JointLimits.acceleration / .jerk: per joint,
    #: kinematics_loader.py:1102-1124, consumed per dof by cost/wp_cspace_state.py:20-287)
    max_acceleration: Union[float, Sequence[float]] = 15.0  # content/configs/robot/franka.yml:48-49
    max_jerk: Union[float, Sequence[float]] = 500.0
    #: joint-torque limits (reference: the c-space STATE cost's effort bound on inverse-dynamics torques,
    #: cost/wp_cspace_state.py + cuda_ops/dynamics.py RNEA; "motion generation with torque limits",
    #: docs/reference/benchmarks.rst:32-42): tau = RNEA(q, qd, qdd) per point, bound cost with
    #: cspace_weight[4] / activation[4], squared regularisation cspace_regularization[3]; the VJP goes back
    #: through the RNEA backward kernel.  Runs on the kernel sequence (the fused launch has no RNEA).
    use_torque_limits: bool = False
    effort_limit: Optional[List[float]] = None  # per joint max |tau|; None = the robot's URDF effort limits
    #: kernel sequence with torque limits: run the joint-space chain (RNEA -> c-space STATE -> RNEA VJP) on a side stream
    #: next to the task-space chain (FK -> costs -> FK VJP); same kernels, same numbers (evaluate_action)
    overlap_dynamics: bool = True
    #: with torque limits the fused launch carries the inverse dynamics of 33 points on 33 lanes of each workgroup: it wins
    #: while the batch fits one round of workgroups (two per CU: 512 rollouts, ~103 us) and loses to the kernel sequence with
    #: its side stream beyond (1024 rollouts: 201 vs 181 us, 4096: 759 vs 548 us; tools/probes/torque_rollout_time.py)
    fused_torque_max_batch: int = 768
    gravity: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0, 0.0, 0.0, 9.81])  # spatial base acceleration
    #: one fused launch (csrc/rollout_fused.hip with the trajopt terms) when a trajectory fits in LDS
    use_fused: bool = True
    longest_first_dispatch: bool = True  # see CollisionRolloutCfg
    # (fields added after round 4 stay at the END: positional construction keeps its meaning)
    #: compile a compile-time shape of the fused launch for this robot / horizon at first use when the library holds none
    #: (``backends/fused_jit.py``; also CUROBO_HIP_JIT_SHAPES=1): see CollisionRolloutCfg.jit_shape
    jit_shape: bool = False
    #: joint-position tracking of the c-space STATE cost (wp_cspace_state.py:205-225): weight of |q - target|^2 at the last point,
    #: times ``cspace_non_terminal_weight_factor`` at the points before it.  0 = off (the trajopt task, lbfgs_bspline_trajopt.yml:72);
    #: the MPC task tracks an IK solution of its pose goal with 1000 / 0.05 (lbfgs_mpc.yml:28-29).  The target itself comes through
    #: ``TrajOptRollout.update_cspace_target``; ``enable_cspace_target`` / ``disable_cspace_target`` switch the term at run time.
    cspace_target_weight: float = 0.0
    cspace_non_terminal_weight_factor: float = 1.0

    @property
    def horizon(self) -> int:
        return (self.n_knots + self.bspline_degree + 1) * self.interpolation_steps

    @property
    def padded_horizon(self) -> int:
        return self.horizon + 1


class TrajOptRollout:
    """cost[B] and d cost / d knots [B, n_knots * D] of B trajectories from one shared start state
    towards per-row goal poses (goal joint state = implicit rest at the last knot)."""

    def __init__(self, kin: KinematicsParams, scene: Optional[SceneData], batch_size: int,
                 cfg: Optional[TrajOptRolloutCfg] = None):
        self.kin, self.scene, self.cfg = kin, scene, cfg or TrajOptRolloutCfg()
        self.device = kin.device
        self.action_horizon, self.action_dim = self.cfg.n_knots, kin.num_dof
        d, T, D, c = self.device, kin.num_pose_links, kin.num_dof, self.cfg
        f = lambda v: torch.tensor(v, device=d, dtype=torch.float32)  # noqa: E731
        self._w_self, self._w_scene = f([c.self_collision_weight]), f([c.scene_collision_weight])
        self._eta_scene, self._speed_dt = f([c.scene_activation_distance]), f([c.traj_dt])
        self._traj_dt, self._implicit_goal = f([c.traj_dt]), torch.zeros(1, dtype=torch.uint8, device=d)
        self._pose_w = f(c.pose_weight)
        self._axes_w, self._axes_w0 = torch.ones(T, 6, device=d), torch.full((T, 6), float(c.non_terminal_pose_factor), device=d)
        self._tol = f([c.pose_convergence_tolerance] * T)
        self._tol0 = self._tol.clone()
        self._project = torch.zeros(T, dtype=torch.uint8, device=d)
        self._cs_w, self._cs_eta, self._cs_reg = f(c.cspace_weight), f(c.cspace_activation_distance), f(c.cspace_regularization)
        self._p_b, self._v_b = kin.joint_limits_position.contiguous(), kin.joint_limits_velocity.contiguous()
        ones = torch.ones(D, device=d)
        amax, jmax = joint_limit_vector(c.max_acceleration, D, d, "max_acceleration"), joint_limit_vector(c.max_jerk, D, d, "max_jerk")
        self._a_b = torch.stack([-amax, amax])
        self._j_b = torch.stack([-jmax, jmax])
        self._effort_b = torch.stack([-1e9 * ones, 1e9 * ones])
        if c.use_torque_limits:
            lim = f(c.effort_limit) if c.effort_limit is not None else kin.joint_limits_effort
            if lim is None:
                raise ValueError("use_torque_limits needs effort_limit (the robot model carries no effort limits)")
            self._effort_b = torch.stack([-lim.abs(), lim.abs()]).contiguous()
            self._gravity = f(c.gravity)
        self._zero1, self._zeroD, self._onesD = torch.zeros(1, device=d), torch.zeros(1, D, device=d), ones
        # joint-position tracking: _cs_target [B, D] holds every trajectory's target (allocated with the batch buffers); weight 0 = off
        self._cs_tw = torch.zeros(1, device=d)  # (off until enable_cspace_target)
        self._cs_nt, self._cs_dofw = f([c.cspace_non_terminal_weight_factor]), torch.ones(D, device=d)
        self.batch_size = 0
        self._fused_ok: Optional[bool] = None
        self._dispatch = None
        self._terms = None
        self.update_batch_size(batch_size)
        self.update_start_state(None)

    def update_batch_size(self, B: int) -> None:
        if B == self.batch_size:
            return
        k, d, c = self.kin, self.device, self.cfg
        H, D, S, L, T = c.padded_horizon, k.num_dof, k.num_spheres, k.num_links, k.num_pose_links
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=d, dtype=dt)  # noqa: E731
        self.batch_size = B
        self._fused_ok = None  # (the choice between the fused launch and the kernel sequence depends on the batch)
        self.position, self.velocity, self.acceleration, self.jerk = z(B, H, D), z(B, H, D), z(B, H, D), z(B, H, D)
        self.out_dt, self.state_dt = z(B), torch.full((B,), c.traj_dt, device=d)
        self.start_idx, self.goal_idx, self.env_query_idx = z(B, dt=torch.int32), z(B, dt=torch.int32), z(B, dt=torch.int32)
        self.link_pos, self.link_quat = z(B, H, T, 3), z(B, H, T, 4)
        self.robot_spheres, self.cumul_mat, self.com = z(B, H, S, 4), z(B, H, L, 3, 4), z(B, H, 4)
        self.pose_cost, self.pose_pos_dist, self.pose_rot_dist = z(B, H, 2 * T), z(B, H, T), z(B, H, T)
        self.pose_grad_pos, self.pose_grad_quat = z(B, H, T, 3), z(B, H, T, 4)
        self.goalset_idx = z(B, H, T, dt=torch.int32)
        self.cspace_cost = z(B, H, D)
        self.cs_gp, self.cs_gv, self.cs_ga, self.cs_gj = z(B, H, D), z(B, H, D), z(B, H, D), z(B, H, D)
        self.self_dist, self.self_grad, self.self_sparse = z(B, H, 1), z(B, H, S, 4), z(B, H, S, dt=torch.uint8)
        self.scene_dist, self.scene_grad = z(B, H, S), z(B, H, S, 4)
        self._pd, self._bbmv, self._bbmi = z(1), z(1), z(2, dt=torch.int16)
        self.point_cost, self.cost = z(B, H, 1), z(B)
        self.grad_q, self.grad_knots = z(B, H, D), z(B, c.n_knots, D)
        self.idxs_goal, self._idx0 = z(B, dt=torch.int32), z(B, dt=torch.int32)
        self._cs_target = z(B, D)  # the target of every trajectory (update_cspace_target gathers its rows here)
        self._cs_target_idx = torch.arange(B, device=d, dtype=torch.int32)
        self.goal_position, self.goal_quat = z(1, T, 1, 3), z(1, T, 1, 4)
        self.goal_quat[..., 0] = 1.0

    def update_start_state(self, start_position: Optional[torch.Tensor], start_velocity: Optional[torch.Tensor] = None,
                           start_acceleration: Optional[torch.Tensor] = None, start_idx: Optional[torch.Tensor] = None) -> None:
        """Start state(s) of the trajectories: position [n, D] (+ velocity / acceleration for a robot in motion: the
        B-spline's fixed knots reproduce them, bspline_boundary_constraint.cuh:330-367); ``start_idx`` [B] picks the
        start state of every trajectory (reference ``idxs_start``; default: state 0)."""
        D, d = self.action_dim, self.device
        if start_position is None:
            start_position = torch.zeros(1, D, device=d)
        sp = start_position.to(d, torch.float32).reshape(-1, D).contiguous()
        if start_idx is not None:
            self.start_idx.copy_(start_idx.to(device=d, dtype=torch.int32).reshape(-1))
        if getattr(self, "start_pos", None) is not None and self.start_pos.shape == sp.shape:
            self.start_pos.copy_(sp)  # keep the pointers a captured hipGraph holds
            self.start_vel.copy_(start_velocity.to(d, torch.float32).reshape(-1, D)) if start_velocity is not None else self.start_vel.zero_()
            self.start_acc.copy_(start_acceleration.to(d, torch.float32).reshape(-1, D)) if start_acceleration is not None else self.start_acc.zero_()
            return
        self.start_pos = sp.clone()
        n = self.start_pos.shape[0]
        self.start_vel, self.start_acc, self.start_jerk = (torch.zeros(n, D, device=d) for _ in range(3))
        if start_velocity is not None:
            self.start_vel.copy_(start_velocity.to(d, torch.float32).reshape(-1, D))
        if start_acceleration is not None:
            self.start_acc.copy_(start_acceleration.to(d, torch.float32).reshape(-1, D))
        if getattr(self, "goal_pos", None) is None:
            self.goal_pos, self.goal_vel, self.goal_acc, self.goal_jerk = (torch.zeros(1, D, device=d) for _ in range(4))

    def update_goal_state(self, goal_joint_position: Optional[torch.Tensor], goal_idx: Optional[torch.Tensor] = None,
                          implicit: bool = True) -> None:
        """Implicit goal state (reference ``use_implicit_goal_state``, bspline_interpolation.cuh:
        110-124): trajectory b ends at rest in joint configuration ``goal_joint_position[goal_idx[b]]``
        (enforced by the spline's goal boundary knots, not by a cost).  ``None`` = free end point that
        only comes to rest (replicated last knot); ``implicit=False`` keeps the goal rows (and their dt, which the
        kernels index with ``goal_idx`` like the reference's ``seed_goal_js.dt``) but leaves the end point free."""
        D, d = self.action_dim, self.device
        if goal_joint_position is None:
            self.goal_pos = torch.zeros(1, D, device=d)
            self.goal_idx.zero_()
            self._implicit_goal = torch.zeros(1, dtype=torch.uint8, device=d)
            self._traj_dt = torch.full((1,), self.cfg.traj_dt, device=d)
            n = 1
        else:
            g = goal_joint_position.to(d, torch.float32).reshape(-1, D).contiguous()
            n = g.shape[0]
            self.goal_idx.copy_(goal_idx.to(torch.int32) if goal_idx is not None else torch.zeros_like(self.goal_idx))
            if self.goal_pos.shape == g.shape and self._implicit_goal.shape[0] == n and self._traj_dt.shape[0] == n:
                self.goal_pos.copy_(g)  # same shape: keep the pointers a captured hipGraph holds
                self._implicit_goal.fill_(1 if implicit else 0)
                return
            self.goal_pos = g.clone()
            self._implicit_goal = torch.full((n,), 1 if implicit else 0, dtype=torch.uint8, device=d)
            self._traj_dt = torch.full((n,), self.cfg.traj_dt, device=d)
        self.goal_vel, self.goal_acc, self.goal_jerk = (torch.zeros(n, D, device=d) for _ in range(3))

    def update_traj_dt(self, dt, speed_dt: Optional[torch.Tensor] = None) -> None:
        """Time step of the trajectories (reference ``RobotRollout.update_goal_dt``, ``seed_goal_js.dt``): a scalar or one
        value per goal-state row (``update_goal_state``); trajectory b runs at ``dt[goal_idx[b]]``.  In place (captured
        graphs keep their pointers): the B-spline's dt, the c-space cost's per-trajectory dt and the speed metric's dt --
        which is element 0 of the per-trajectory vector, as the reference's kernel reads it (wp_speed_metric.py:54)."""
        if torch.is_tensor(dt):
            self._traj_dt.copy_(dt.to(self.device, torch.float32).reshape(-1))
        else:
            self._traj_dt.fill_(float(dt))
        if self._traj_dt.shape[0] == 1:
            self.state_dt.copy_(self._traj_dt.expand(self.batch_size))
        else:
            torch.index_select(self._traj_dt, 0, self.goal_idx.long(), out=self.state_dt)
        # (a seed shard passes the dt of GLOBAL trajectory 0, which lives on rank 0, so that every world size computes
        # the speed metric the single-process solve does)
        self._speed_dt.copy_(self.state_dt[:1] if speed_dt is None else speed_dt.reshape(1))

    def compute_state_from_action(self, act_seq: torch.Tensor) -> None:
        """knots -> position / velocity / acceleration / jerk buffers only (reference ``compute_state_from_action``)"""
        c, B = self.cfg, self.batch_size
        trajectory_hip.launch_bspline_interpolation_forward_kernel(
            self.position, self.velocity, self.acceleration, self.jerk, self.out_dt, act_seq, self.start_pos,
            self.start_vel, self.start_acc, self.start_jerk, self.goal_pos, self.goal_vel, self.goal_acc, self.goal_jerk,
            self.start_idx, self.goal_idx, self._traj_dt, self._implicit_goal, B, c.padded_horizon, self.action_dim, c.n_knots,
            c.bspline_degree)

    use_multi_env = False

    def update_env_query_idx(self, env_query_idx: Optional[torch.Tensor]) -> None:
        """Scene environment of every trajectory (reference ``idxs_env`` / ``use_multi_env``,
        cost/cost_scene_collision.py:58-198; batch-env planning, motion_planner_batch.py):
        trajectory b collides with the obstacles of environment ``env_query_idx[b]``; ``None`` = env 0.
        Switching between ``None`` and indices changes a kernel argument: re-capture graphs after it."""
        self.use_multi_env = env_query_idx is not None
        if env_query_idx is None:
            self.env_query_idx.zero_()
        else:
            validate_env_query_idx(env_query_idx, self.scene, self.kin.num_envs)
            self.env_query_idx.copy_(env_query_idx.to(device=self.device, dtype=torch.int32).reshape(-1))

    def update_goals(self, goal_position: torch.Tensor, goal_quat: torch.Tensor, idxs_goal: torch.Tensor) -> None:
        """goal_position [G, T, num_goalset, 3], goal_quat (wxyz) [G, T, num_goalset, 4], idxs_goal [B]: every point is
        scored against the closest member of its row's goal set (reference goal sets, cost/tool_pose kernels)."""
        if goal_position.shape == self.goal_position.shape:
            self.goal_position.copy_(goal_position)
            self.goal_quat.copy_(goal_quat)
        else:  # new buffers (and a new goal-set size): the fused launch's argument block is rebuilt on the next evaluation
            self.goal_position = goal_position.to(self.device, torch.float32).contiguous().clone()
            self.goal_quat = goal_quat.to(self.device, torch.float32).contiguous().clone()
            self._terms = None
        self.idxs_goal.copy_(idxs_goal.to(torch.int32))

    def update_cspace_target(self, target_position: torch.Tensor, idxs_target: Optional[torch.Tensor] = None,
                             dof_weight: Optional[torch.Tensor] = None) -> None:
        """joint-position tracking target of the c-space STATE cost: ``target_position`` [G, D], trajectory b tracks row
        ``idxs_target[b]`` (default: its pose goal's row, ``idxs_goal``); ``dof_weight`` [D] scales the joints.  Same shapes are
        written in place (captured graphs see them); the term counts once ``enable_cspace_target`` gave it a weight"""
        t = target_position.to(self.device, torch.float32).reshape(-1, self.action_dim)
        idx = (self.idxs_goal if idxs_target is None else idxs_target).to(self.device, torch.int64).reshape(-1)
        if int(idx.max()) >= t.shape[0] or int(idx.min()) < 0:
            raise ValueError(f"idxs_target outside the {t.shape[0]} target rows")
        # one row per trajectory, written in place: the buffer never moves, so launches captured in a hipGraph before the
        # first target was given read the new values too
        self._cs_target.copy_(t[idx])
        if dof_weight is not None:
            self._cs_dofw.copy_(dof_weight.to(self.device, torch.float32).reshape(-1))

    def enable_cspace_target(self, weight: Optional[float] = None, non_terminal_weight_factor: Optional[float] = None) -> None:
        """switch joint-position tracking on (reference CSpaceCost.enable_cspace_target): the configured weight / factor, or these"""
        c = self.cfg
        w = c.cspace_target_weight if weight is None else float(weight)
        if w <= 0.0:
            raise ValueError("enable_cspace_target needs a positive weight (cfg.cspace_target_weight or the argument)")
        self._cs_tw.fill_(w)
        self._cs_nt.fill_(c.cspace_non_terminal_weight_factor if non_terminal_weight_factor is None else float(non_terminal_weight_factor))

    def disable_cspace_target(self) -> None:
        self._cs_tw.zero_()

    def update_tool_pose_criteria(self, criteria) -> None:
        """``{tool frame: ToolPoseCriteria}`` -> the per-frame factor / tolerance / projection rows the pose cost reads
        (reference ToolPoseCost.update_tool_pose_criteria); written in place, so captured graphs see the new values"""
        for name, c in criteria.items():
            if name not in self.kin.tool_frames:
                raise ValueError(f"tool frame {name} not in {self.kin.tool_frames}")
            i, f = self.kin.tool_frames.index(name), lambda v: torch.tensor(v, device=self.device, dtype=torch.float32)  # noqa: E731
            self._axes_w[i].copy_(f(c.terminal_pose_axes_weight_factor))
            self._axes_w0[i].copy_(f(c.non_terminal_pose_axes_weight_factor))
            self._tol[i].copy_(f(c.terminal_pose_convergence_tolerance))
            self._tol0[i].copy_(f(c.non_terminal_pose_convergence_tolerance))
            self._project[i] = int(bool(c.project_distance_to_goal))

    # ------------------------------------------------------------------ forward + backward
    def evaluate_action(self, act_seq: torch.Tensor, with_gradient: bool = True) -> torch.Tensor:
        k, c, B = self.kin, self.cfg, self.batch_size
        H, D, S, T = c.padded_horizon, k.num_dof, k.num_spheres, k.num_pose_links
        trajectory_hip.launch_bspline_interpolation_forward_kernel(
            self.position, self.velocity, self.acceleration, self.jerk, self.out_dt, act_seq, self.start_pos,
            self.start_vel, self.start_acc, self.start_jerk, self.goal_pos, self.goal_vel, self.goal_acc, self.goal_jerk,
            self.start_idx, self.goal_idx, self._traj_dt, self._implicit_goal, B, H, D, c.n_knots, c.bspline_degree)
        tq = c.use_torque_limits
        # The joint-space chain (inverse dynamics -> c-space STATE cost -> RNEA VJP) reads the B-spline samples only, the
        # task-space chain (FK -> tool pose -> self / scene collision -> FK VJP) the joint positions only: with torque
        # limits on, the first runs on a side stream next to the second and the two meet at the per-point aggregate.  The
        # serial tree walks of RNEA occupy half a wavefront per SIMD (528 wavefronts for 33 792 elements) and leave the
        # chip to the collision kernels; one after the other they cost the sum (Unitree G1, C4 shapes: 1 428 us).
        side = None
        if tq and c.overlap_dynamics and self.position.is_cuda and not inside_forked_stream():
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
            side = self._side_stream
            if getattr(self, "_tau", None) is None or self._tau.shape[0] != B * H:
                side = None  # first call: the buffers below are allocated on the calling stream
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            if tq:  # inverse dynamics of every trajectory point (reference cuda_ops/dynamics.py, RNEA forward)
                n, L = B * H, k.num_links
                if getattr(self, "_tau", None) is None or self._tau.shape[0] != n:
                    z = lambda *s: torch.zeros(*s, device=self.device)  # noqa: E731
                    self._tau, self._rnea_cache, self._rnea_ws = z(n, D), z(n, L * 20), z(n, L * 18)
                    self._cs_gtau, self._rnea_g = z(B, H, D), [z(n, D) for _ in range(3)]
                    # with the joint-space chain on the side stream the walks read their inputs from a transposed scratch
                    # instead of staging them through LDS: the CU's LDS stays with the collision kernels they run next to
                    # (they walk with an element per lane, a quarter of the staged quad walk's wavefronts: next to other kernels that
                    # wins for a 7-dof arm as well -- Franka, 1024 / 4096 rollouts: 160 / 488 us against 177 / 543)
                    # (below ~24 k points the two extra transposition launches cost more than the walks gain: 512 rollouts 139 / 130 us)
                    self._rnea_scratch = z(3 * n * D) if (c.overlap_dynamics and self.position.is_cuda and (D > 20 or n >= 24576)) else None
                rargs = (k.fixed_transforms, k.link_masses_com, k.link_inertias, k.joint_map_type, k.joint_map, k.link_map,
                         k.joint_offset_map, self._gravity, k.link_level_offsets, k.link_level_data)
                dynamics_hip.launch_rnea_forward(self._tau, self.position.view(n, D), self.velocity.view(n, D),
                                                 self.acceleration.view(n, D), *rargs, self._rnea_cache, n, L, D, k.n_tree_levels, 1, None,
                                                 scratch=self._rnea_scratch)
            cost_hip.cspace_state_cost(
                self.cspace_cost, self.cs_gp, self.cs_gv, self.cs_ga, self.cs_gj, self._cs_gtau if tq else None, self.position,
                self.velocity, self.acceleration, self.jerk, self._tau.view(B, H, D) if tq else None, self.state_dt, self._cs_target,
                self._cs_target_idx, self._p_b, self._v_b, self._a_b, self._j_b, self._effort_b, self._cs_w, self._cs_eta, self._cs_reg,
                self._cs_tw, self._cs_nt, self._cs_dofw, True, B, H, D, c.retime_weights, c.retime_regularization_weights)
            if tq and with_gradient:  # d cost / d tau back to (q, qd, qdd): RNEA VJP, added to the c-space gradients
                if self._rnea_scratch is not None:  # straight into the c-space gradients (no separate buffers, no adds)
                    dynamics_hip.launch_rnea_backward(self.cs_gp.view(n, D), self.cs_gv.view(n, D), self.cs_ga.view(n, D),
                                                      self._cs_gtau.view(n, D), self.position.view(n, D), self.velocity.view(n, D), *rargs,
                                                      self._rnea_cache, n, L, D, k.n_tree_levels, 1, None, self._rnea_ws,
                                                      scratch=self._rnea_scratch, scratch_holds_q_qd=True, accumulate=True)
                else:
                    dynamics_hip.launch_rnea_backward(*self._rnea_g, self._cs_gtau.view(n, D), self.position.view(n, D),
                                                      self.velocity.view(n, D), *rargs, self._rnea_cache, n, L, D, k.n_tree_levels, 1,
                                                      None, self._rnea_ws)
                    self.cs_gp.view(n, D).add_(self._rnea_g[0])
                    self.cs_gv.view(n, D).add_(self._rnea_g[1])
                    self.cs_ga.view(n, D).add_(self._rnea_g[2])
        # (the task-space chain is enqueued AFTER the joint-space chain: the few, long-running workgroups of the tree walks
        # take their slots on an empty chip; started behind the collision kernels they wait for LDS that those keep taking)
        kinematics_hip.launch_kinematics_forward_spheres(
            self.link_pos, self.link_quat, self.robot_spheres, self.com, self.cumul_mat, self.position,
            k.fixed_transforms, k.link_spheres, k.link_masses_com, k.joint_map_type, k.joint_map, k.link_map,
            k.tool_frame_map, k.link_sphere_idx_map, k.joint_offset_map, self.env_query_idx, k.num_envs, B * H, H, D, S,
            32, True, False)
        cost_hip.tool_pose_distance(
            self.pose_cost, self.pose_pos_dist, self.pose_rot_dist, self.pose_grad_pos, self.pose_grad_quat,
            self.goalset_idx, self.link_pos, self.link_quat, self.goal_position, self.goal_quat, self.idxs_goal,
            self._pose_w, self._axes_w, self._axes_w0, self._tol, self._tol0, self._project, B, H, T, int(self.goal_position.shape[2]),
            c.rotation_method)
        sc = k.self_collision
        geometry_hip.self_collision_distance(
            self.self_dist, self.self_grad, self._pd, self.self_sparse, self.robot_spheres, sc.sphere_padding,
            self._w_self, sc.collision_pairs, self._bbmv, self._bbmi, 1, 256, B, H, S, sc.collision_pairs.shape[0],
            False, True)
        use_scene = self.scene is not None
        if use_scene:
            collision_hip.sphere_obstacle_collision(
                self.scene_dist, self.scene_grad, self.robot_spheres, self.scene.struct, self._w_scene, self._eta_scene,
                self.env_query_idx, B, H, S, self.use_multi_env, 3 if c.use_sweep else 0, c.use_sweep and c.use_speed_metric,
                self._speed_dt)
        if with_gradient:
            kinematics_hip.launch_kinematics_backward(
                self.grad_q, self.pose_grad_pos, self.pose_grad_quat, self.self_grad, self.com, self.com,
                self.pose_grad_pos, self.cumul_mat, k.link_spheres, k.link_masses_com, k.link_map, k.joint_map,
                k.joint_map_type, k.tool_frame_map, k.link_sphere_idx_map, k.link_chain_data, k.link_chain_offsets,
                k.joint_links_data, k.joint_links_offsets, k.joint_affects_endeffector, k.joint_offset_map,
                self.env_query_idx, k.num_envs, B * H, H, D, S, False, False,
                grad_spheres_b=self.scene_grad if use_scene else None)
        if side is not None:
            torch.cuda.current_stream(self.device).wait_stream(side)
        # per-point totals (+ the c-space position gradient into grad_q), then the sum over the horizon
        cost_hip.rollout_point_aggregate(
            self.point_cost, self.grad_q if with_gradient else None, self.pose_cost, self.cspace_cost,
            self.cs_gp if with_gradient else None, self.self_dist, self.scene_dist if use_scene else None, B * H, T, D, S)
        collision_hip.trajectory_cost_sum(self.cost, None, self.point_cost, B, H, 1)
        if with_gradient:
            trajectory_hip.launch_bspline_interpolation_backward_kernel(
                self.grad_knots, self.grad_q, self.cs_gv, self.cs_ga, self.cs_gj, self._traj_dt, self.goal_idx,
                self._implicit_goal, B, H, D, c.n_knots, c.bspline_degree, False)
        return self.cost

    # ------------------------------------------------------------------ fused
    def fused_available(self) -> bool:
        k, c = self.kin, self.cfg
        n_obs = (self.scene.struct.max_cuboids + self.scene.struct.max_voxel_grids) if self.scene is not None else 0
        need = rollout_hip.rollout_trajopt_fused_lds_bytes(
            c.padded_horizon, k.num_dof, k.num_links, k.num_spheres, int(k.self_collision.collision_pairs.shape[0]),
            int(k.link_chain_data.shape[0]), n_obs, True)
        ok = need <= rollout_hip.FUSED_LDS_LIMIT and k.num_links <= 128 and k.num_dof <= 64
        if self.scene is not None and getattr(self.scene.struct, "mesh_set", None) is not None:
            ok = False  # mesh obstacles are queried by their own launch (BVH): the kernel sequence runs
        if ok and c.use_torque_limits:  # inverse dynamics inside the launch borrows LDS regions that are dead by then
            ok = rollout_hip.rollout_trajopt_fused_torque_fits(
                c.padded_horizon, k.num_dof, k.num_links, k.num_spheres, int(k.self_collision.collision_pairs.shape[0]),
                int(k.link_chain_data.shape[0]), n_obs)
        return ok

    def _maybe_jit_shape(self) -> None:
        from ..backends import fused_jit

        if not (self.cfg.jit_shape or fused_jit.enabled_by_env()):
            return
        k, c = self.kin, self.cfg
        lanes = getattr(k.self_collision.collision_pairs, "_self_lane_lists", None)
        n_obs = (self.scene.struct.max_cuboids + self.scene.struct.max_voxel_grids) if self.scene is not None else 0
        fused_jit.ensure_shape(c.padded_horizon, c.n_knots, k.num_dof, k.num_links, k.num_spheres,
                               int(k.self_collision.collision_pairs.shape[0]), int(k.link_chain_data.shape[0]),
                               int(lanes[1]) if lanes is not None else 0, n_obs, with_trajopt_terms=True)

    def _dispatch_order(self):
        if not self.cfg.longest_first_dispatch:
            return None
        if self._dispatch is None:
            self._dispatch = rollout_hip.DispatchOrder(self.batch_size, self.cost.device)
        return self._dispatch

    def cost_and_gradient_fused(self, act_seq: torch.Tensor, with_metrics: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """Same numbers as ``evaluate_action`` from one launch (the struct of optional terms is
        rebuilt per call: it only holds pointers of the static buffers)."""
        k, c, B = self.kin, self.cfg, self.batch_size
        sc, m = k.self_collision, with_metrics
        use_scene = self.scene is not None
        if c.use_torque_limits:  # the terms struct holds the raw pointer of the walk order: keep the tensor referenced
            self._level_links = dynamics_hip._walk_order(k.link_map, k.link_level_data)
        self._terms = rollout_hip.make_trajopt_terms(
            out_pose_distance=self.pose_cost if m else None, out_position_distance=self.pose_pos_dist if m else None,
            out_rotation_distance=self.pose_rot_dist if m else None, out_goalset_idx=self.goalset_idx if m else None,
            goal_position=self.goal_position, goal_quat=self.goal_quat, idxs_goal=self.idxs_goal,
            position_orientation_weight=self._pose_w, terminal_pose_axes_weight_factor=self._axes_w,
            non_terminal_pose_axes_weight_factor=self._axes_w0, terminal_pose_convergence_tolerance=self._tol,
            non_terminal_pose_convergence_tolerance=self._tol0, project_distance_to_goal=self._project,
            tool_frame_map=k.tool_frame_map, n_tool_frames=k.num_pose_links, num_goalset=int(self.goal_position.shape[2]),
            rotation_method=c.rotation_method, out_cspace_cost=self.cspace_cost if m else None, state_dt=self.state_dt,
            target_joint_position=self._cs_target, idxs_target_joint_position=self._cs_target_idx, p_b=self._p_b, v_b=self._v_b,
            a_b=self._a_b, j_b=self._j_b, effort_b=self._effort_b, cspace_weight=self._cs_w,
            cspace_activation_distance=self._cs_eta, squared_l2_regularization_weights=self._cs_reg,
            cspace_target_weight=self._cs_tw, cspace_non_terminal_weight_factor=self._cs_nt,
            cspace_target_dof_weight=self._cs_dofw, retime_weights=c.retime_weights,
            retime_regularization_weights=c.retime_regularization_weights,
            **(dict(link_masses_com=k.link_masses_com, link_inertias=k.link_inertias, gravity=self._gravity,
                    level_links=self._level_links, use_torque_limits=1)
               if c.use_torque_limits else {}))
        rollout_hip.rollout_trajopt_fused(
            self._terms, self.cost, self.grad_knots, self.position if m else None, self.robot_spheres if m else None,
            act_seq, self.start_pos, self.start_vel, self.start_acc, self.start_jerk, self.goal_pos, self.goal_vel,
            self.goal_acc, self.goal_jerk, self.start_idx, self.goal_idx, self._traj_dt, self._implicit_goal,
            k.fixed_transforms, k.link_spheres, k.joint_map_type, k.joint_map, k.link_map, k.link_sphere_idx_map,
            k.link_chain_data, k.link_chain_offsets, k.joint_offset_map, sc.sphere_padding, self._w_self,
            sc.collision_pairs, self.scene.struct if use_scene else None, self._w_scene if use_scene else None,
            self._eta_scene, self._speed_dt, self.env_query_idx, k.num_envs, self.use_multi_env, B, c.padded_horizon, self.action_dim,
            c.n_knots, c.bspline_degree, 3 if c.use_sweep else 0, c.use_sweep and c.use_speed_metric,
            dispatch=self._dispatch_order())
        return self.cost, self.grad_knots.view(B, -1)

    def cost_and_gradient(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        act = x.view(self.batch_size, self.cfg.n_knots, self.action_dim)
        if self.cfg.use_fused:
            if self._fused_ok is None:
                c = self.cfg
                too_big = c.use_torque_limits and c.overlap_dynamics and self.batch_size > c.fused_torque_max_batch
                self._fused_ok = self.fused_available() and not too_big
                if self._fused_ok:
                    self._maybe_jit_shape()
            if self._fused_ok:
                return self.cost_and_gradient_fused(act)
        cost = self.evaluate_action(act, with_gradient=True)
        return cost, self.grad_knots.view(self.batch_size, -1)
