"""``TrajectoryOptimizer``, ``MotionPlanner`` and ``BatchMotionPlanner``: the reference's planning front ends
over the HIP solvers.

* ``TrajectoryOptimizer`` / ``TrajectoryOptimizerCfg`` / ``TrajectoryOptimizerResult`` = the reference's
  ``curobo.trajectory_optimizer`` (``TrajOptSolver`` / ``TrajOptSolverCfg.create`` / ``TrajOptSolverResult``,
  ``curobo/_src/solver/solver_trajopt.py``, ``solver_trajopt_cfg.py:118-240``) with ``JointState`` /
  ``GoalToolPose`` arguments.
* ``MotionPlanner.plan_pose`` = ``curobo/_src/motion/motion_planner.py:207-296``: per attempt, collision-free
  IK with ``return_seeds = num_trajopt_seeds`` -> failed solutions replaced by the first good one -> trajectory
  optimisation seeded with straight lines to those solutions, implicit goal state, one time-optimal finetune
  pass (dt scale 0.55) -> stop at the first attempt with a successful seed.  ``plan_cspace`` (:329-396): joint-space
  goal, three finetune passes at 0.75.  The PRM graph planner that seeds later attempts in the reference
  (``_get_graph_seed_trajectories``) is out of scope (graph search), so every attempt is the
  IK-seeded one.  ``plan_grasp`` (:419-588): goal-set plan to the grasp candidates -> approach pose -> straight-line motion
  to the chosen grasp -> straight-line lift, with the grasp-contact links' collision spheres switched off where the
  reference switches them off.  The attachment manager is not mirrored.
* ``BatchMotionPlanner.plan_pose`` / ``plan_cspace`` = ``motion_planner_batch.py:139-289``: ``max_batch_size``
  problems in one IK + trajopt pass, per-problem start states, first-success-wins over the attempts, optional
  one-world-per-problem (``multi_env``: problem p collides with scene environment p).

With ``torch.distributed`` initialised the solvers shard their seeds over the ranks (``IKSolver.sharded`` /
``TrajOptSolver.sharded``); the planners' host decisions use results that are identical on every rank.
"""

from __future__ import annotations

import os
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

import numpy as np
import torch

from .kinematics import Kinematics, KinematicsCfg, KinematicsState
from .scene import SceneData
from .scene.config import scene_from_config
from .solver.ik import IKSolver, IKSolverCfg
from .solver.trajopt import TrajOptResult, TrajOptSolver, TrajOptSolverCfg
from .types import DeviceCfg, GoalToolPose, JointState, Pose, ToolPoseCriteria
from .solver.tracking import ToolPoseTrackingMixin


def _load_kinematics(robot, device, assets_root: str = "", num_envs: int = 1) -> KinematicsCfg:
    import os

    if isinstance(robot, KinematicsCfg):
        return robot
    if isinstance(robot, dict):
        return KinematicsCfg.from_data_dict(robot, assets_root=assets_root, device=device, num_envs=num_envs)
    if os.path.exists(str(robot)):
        return KinematicsCfg.from_robot_yaml_file(robot, assets_root or os.path.dirname(os.path.abspath(robot)), device=device,
                                                  num_envs=num_envs)
    return KinematicsCfg.from_packaged(str(robot).replace(".yml", "").replace(".yaml", ""), device=device)


@dataclass
class TrajectoryOptimizerResult:
    """reference TrajOptSolverResult fields (solver_trajopt_result.py:27-60): one row per problem and returned seed"""

    success: torch.Tensor                 # [batch, return_seeds] bool
    solution: torch.Tensor                # [batch, return_seeds, n_knots, dof] optimised knots
    js_solution: JointState               # [batch, return_seeds, horizon, dof] with velocity / acceleration / jerk and dt [batch, return_seeds]
    position_error: torch.Tensor          # [batch, return_seeds]
    rotation_error: torch.Tensor
    interpolated_trajectory: Optional[JointState] = None   # [batch, return_seeds, steps, dof] at interpolation_dt
    interpolated_last_tstep: Optional[torch.Tensor] = None  # [batch, return_seeds]
    seed_cost: Optional[torch.Tensor] = None
    goalset_index: Optional[torch.Tensor] = None
    solve_time: float = 0.0
    total_time: float = 0.0
    debug_info: Optional[dict] = None

    @property
    def motion_time(self) -> torch.Tensor:
        """(horizon - 1) x dt per returned trajectory"""
        return (self.js_solution.position.shape[-2] - 1) * self.js_solution.dt

    def get_interpolated_plan(self) -> JointState:
        """the best trajectory of a single-problem result, trimmed to its last step (reference :133-141)"""
        if self.interpolated_trajectory is None:
            raise ValueError("the result carries no interpolated trajectory")
        if self.interpolated_last_tstep.numel() > 1 and self.interpolated_last_tstep.shape[0] > 1:
            raise ValueError("only single result is supported")
        n = int(self.interpolated_last_tstep.reshape(-1)[0])
        t = self.interpolated_trajectory
        g = lambda x: None if x is None else x.reshape(-1, x.shape[-2], x.shape[-1])[0, :n]  # noqa: E731
        return JointState(g(t.position), g(t.velocity), g(t.acceleration), g(t.jerk), t.joint_names, t.dt)

    def copy_at_batch_indices(self, other: "TrajectoryOptimizerResult", mask: torch.Tensor) -> None:
        """rows of ``other`` where ``mask`` [batch] is set replace this result's (reference copy_at_batch_indices)"""
        def put(a, b):
            if a is None or b is None:
                return a
            if a.shape != b.shape:  # interpolation buffers of different lengths: pad the shorter one with its last sample
                n = max(a.shape[2], b.shape[2])
                pad = lambda x: torch.cat([x, x[:, :, -1:].expand(-1, -1, n - x.shape[2], -1)], 2) if x.shape[2] < n else x  # noqa: E731
                a, b = pad(a), pad(b)
            m = mask.view(-1, *([1] * (a.ndim - 1)))
            return torch.where(m, b, a)
        for name in ("success", "solution", "position_error", "rotation_error", "seed_cost", "interpolated_last_tstep", "goalset_index"):
            setattr(self, name, put(getattr(self, name), getattr(other, name)))
        for js_name in ("js_solution", "interpolated_trajectory"):
            a, b = getattr(self, js_name), getattr(other, js_name)
            if a is None or b is None:
                continue
            for f in ("position", "velocity", "acceleration", "jerk", "dt"):
                setattr(a, f, put(getattr(a, f), getattr(b, f)))


@dataclass
class GraspPlanResult:
    """reference GraspPlanResult (motion/motion_planner_result.py): the three legs of a grasp plan"""

    success: Optional[torch.Tensor] = None
    approach_success: Optional[torch.Tensor] = None
    grasp_success: Optional[torch.Tensor] = None
    lift_success: Optional[torch.Tensor] = None
    status: Optional[str] = None
    goalset_result: Optional[TrajectoryOptimizerResult] = None
    goalset_index: Optional[torch.Tensor] = None
    approach_result: Optional[TrajectoryOptimizerResult] = None
    approach_trajectory: Optional[JointState] = None
    approach_trajectory_dt: Optional[torch.Tensor] = None
    approach_interpolated_trajectory: Optional[JointState] = None
    approach_interpolated_last_tstep: Optional[torch.Tensor] = None
    grasp_trajectory: Optional[JointState] = None
    grasp_trajectory_dt: Optional[torch.Tensor] = None
    grasp_interpolated_trajectory: Optional[JointState] = None
    grasp_interpolated_last_tstep: Optional[torch.Tensor] = None
    lift_trajectory: Optional[JointState] = None
    lift_trajectory_dt: Optional[torch.Tensor] = None
    lift_interpolated_trajectory: Optional[JointState] = None
    lift_interpolated_last_tstep: Optional[torch.Tensor] = None


def _axis_offset_pose(axis: str, offset: float) -> Pose:
    if axis not in ("x", "y", "z"):
        raise ValueError(f"Invalid axis: {axis}, must be 'x', 'y', or 'z'")
    return Pose.from_list([offset * (axis == a) for a in ("x", "y", "z")] + [1.0, 0.0, 0.0, 0.0])


def robot_acceleration_jerk_limits(kinematics: Optional[KinematicsCfg]):
    """(max_acceleration, max_jerk) of the robot file's cspace block, per ACTIVE joint as the loader carries them (reference:
    JointLimits.acceleration / .jerk, which its rollouts bound the trajectory by per joint, kinematics_loader.py:1102-1124,
    cost/wp_cspace_state.py:20-287): a float when every joint has the same value, else the list [dof]; ``None`` for a limit
    the model does not carry (the configuration's default stays)."""
    cs = getattr(getattr(kinematics, "model", None), "cspace", None) or {}
    dof = getattr(getattr(kinematics, "model", None), "num_dof", None)
    out = []
    for key in ("max_acceleration", "max_jerk"):
        v = cs.get(key)
        v = [] if v is None else [float(x) for x in np.atleast_1d(np.asarray(v, np.float64))]
        if not v:
            out.append(None)
        elif all(abs(x - v[0]) <= 1e-9 * max(1.0, abs(v[0])) for x in v):
            out.append(v[0])
        elif dof is not None and len(v) == int(dof):
            out.append(v)
        else:  # a list that is not per active joint cannot be applied joint by joint: the tightest value bounds every joint
            out.append(min(v))
    return tuple(out)


def apply_robot_limits(rollout_cfg, kinematics: Optional[KinematicsCfg]) -> None:
    acc, jerk = robot_acceleration_jerk_limits(kinematics)
    if acc is not None:
        rollout_cfg.max_acceleration = acc
    if jerk is not None:
        rollout_cfg.max_jerk = jerk


@dataclass
class TrajectoryOptimizerCfg:
    kinematics: KinematicsCfg = None
    scene: Optional[SceneData] = None
    device_cfg: DeviceCfg = field(default_factory=DeviceCfg)
    num_seeds: int = 4
    position_tolerance: float = 0.005
    orientation_tolerance: float = 0.05
    use_cuda_graph: bool = True
    self_collision_check: bool = True
    optimizer_collision_activation_distance: float = 0.01
    interpolation_dt: float = 0.025
    minimum_trajectory_dt: float = 0.002
    maximum_trajectory_dt: float = 0.2
    max_batch_size: int = 1
    multi_env: bool = False
    random_seed: int = 123
    num_ik_seeds: int = 32
    #: spline of the optimiser: content/configs/task/trajopt/transition_bspline_trajopt.yml:9-11 in the reference
    #: (n_knots 16, interpolation_steps 4); this backend's default is the C2 shape, 12 knots x 2 = 32 + 1 points
    n_knots: int = 12
    interpolation_steps: int = 2
    #: goal poses per problem the solvers are built for (reference ``max_goalset``); smaller sets are padded
    max_goalset: int = 1

    @staticmethod
    def create(robot: Union[str, Dict, KinematicsCfg], scene_model: Union[str, Dict, List, None] = None, num_seeds: int = 4,
               position_tolerance: float = 0.005, orientation_tolerance: float = 0.05, use_cuda_graph: bool = True,
               self_collision_check: bool = True, optimizer_collision_activation_distance: float = 0.01,
               device_cfg: Optional[DeviceCfg] = None, interpolation_dt: float = 0.025, minimum_trajectory_dt: float = 0.002,
               maximum_trajectory_dt: float = 0.2, max_batch_size: int = 1, multi_env: bool = False, random_seed: int = 123,
               num_ik_seeds: int = 32, assets_root: str = "", n_knots: int = 12, interpolation_steps: int = 2,
               max_goalset: int = 1, **unused) -> "TrajectoryOptimizerCfg":
        """Arguments of the reference's ``TrajOptSolverCfg.create`` (solver_trajopt_cfg.py:118-240).  ``robot``: packaged
        name (``"franka.yml"``), a robot yaml path, its dictionary or a ``KinematicsCfg``; ``scene_model``: the
        reference's scene format (a list = one world per environment).  Keyword arguments this backend has no use
        for (``optimizer_configs``, ``transition_model``, ``metrics_rollout``, ...) are accepted and ignored: the cost set
        and optimiser settings of ``content/configs/task/trajopt/lbfgs_bspline_trajopt.yml`` are built in."""
        device_cfg = device_cfg or DeviceCfg()
        dev = device_cfg.device
        scene = scene_from_config(scene_model, dev, cache=unused.get("collision_cache"))  # (slots reserved for add_obstacle)
        kin = _load_kinematics(robot, dev, assets_root)
        return TrajectoryOptimizerCfg(
            kinematics=kin, scene=scene, device_cfg=device_cfg, num_seeds=num_seeds, position_tolerance=position_tolerance,
            orientation_tolerance=orientation_tolerance, use_cuda_graph=use_cuda_graph, self_collision_check=self_collision_check,
            optimizer_collision_activation_distance=optimizer_collision_activation_distance, interpolation_dt=interpolation_dt,
            minimum_trajectory_dt=minimum_trajectory_dt, maximum_trajectory_dt=maximum_trajectory_dt,
            max_batch_size=max_batch_size, multi_env=multi_env, random_seed=random_seed, num_ik_seeds=num_ik_seeds,
            n_knots=n_knots, interpolation_steps=interpolation_steps, max_goalset=max_goalset)

    def solver_cfg(self) -> TrajOptSolverCfg:
        c = TrajOptSolverCfg(num_seeds=self.num_seeds, position_threshold=self.position_tolerance,
                             rotation_threshold=self.orientation_tolerance, seed=self.random_seed,
                             interpolation_dt=self.interpolation_dt, minimum_trajectory_dt=self.minimum_trajectory_dt,
                             maximum_trajectory_dt=self.maximum_trajectory_dt, num_goalset=self.max_goalset)
        c.rollout.n_knots, c.rollout.interpolation_steps = self.n_knots, self.interpolation_steps
        c.rollout.scene_activation_distance = self.optimizer_collision_activation_distance
        apply_robot_limits(c.rollout, self.kinematics)  # (the UR10e file: 12 rad/s^2, not the Franka's 15)
        if not self.self_collision_check:
            c.rollout.self_collision_weight = 0.0
        c.ik = IKSolverCfg(num_seeds=self.num_ik_seeds, position_threshold=self.position_tolerance,
                           rotation_threshold=self.orientation_tolerance, seed=self.random_seed, num_goalset=self.max_goalset)
        c.ik.rollout.scene_activation_distance = self.optimizer_collision_activation_distance
        return c


class TrajectoryOptimizer(ToolPoseTrackingMixin):
    """the reference's ``TrajOptSolver`` call surface: ``solve_pose(goal_tool_poses, current_state, ...)`` and
    ``solve_cspace(goal_state, current_state, ...)`` on ``max_batch_size`` problems (smaller batches are padded with
    their first problem, as the reference does, :759-775)"""

    def __init__(self, config: TrajectoryOptimizerCfg):
        self.config = config
        self.kinematics = Kinematics(config.kinematics, compute_spheres=True)
        self._solver: Optional[TrajOptSolver] = None
        self.solve_time = 0.0

    # ---- reference members
    @property
    def joint_names(self) -> List[str]:
        return self.kinematics.joint_names

    @property
    def tool_frames(self) -> List[str]:
        return self.kinematics.tool_frames

    @property
    def action_dim(self) -> int:
        return self.config.kinematics.kinematics_config.num_dof

    @property
    def action_horizon(self) -> int:
        return self.config.n_knots

    @property
    def interpolation_steps(self) -> int:
        return self.config.interpolation_steps

    @property
    def default_joint_state(self) -> JointState:
        from .workloads import start_configuration

        q = torch.as_tensor(start_configuration(self.config.kinematics.model), device=self.config.device_cfg.device)
        return JointState.from_position(q, joint_names=self.joint_names)

    def compute_kinematics(self, state: Union[JointState, torch.Tensor]) -> KinematicsState:
        return self.kinematics.compute_kinematics(state)

    def update_world(self, scene) -> None:
        """``SceneData``, or a scene description (``curobo.scene.Scene``, dictionary, yaml path, list per environment)"""
        self.config.scene = scene if (scene is None or isinstance(scene, SceneData)) else scene_from_config(scene, self.config.device_cfg.device)
        self._solver = None

    def reset_seed(self) -> None:
        if self._solver is not None:
            self._solver.reset_seed()

    def update_tool_pose_criteria(self, tool_pose_criteria: Dict[str, ToolPoseCriteria]) -> None:
        self._criteria = dict(tool_pose_criteria)
        if self._solver is not None:
            self._solver.update_tool_pose_criteria(tool_pose_criteria)

    # ---- further reference members (solver_trajopt.py:121-215, solver_core.py)
    @property
    def device_cfg(self) -> DeviceCfg:
        return self.config.device_cfg

    @property
    def default_joint_position(self) -> torch.Tensor:
        return self.kinematics.default_joint_position

    @property
    def horizon(self) -> int:
        """points of an optimised trajectory"""
        return self.config.solver_cfg().rollout.padded_horizon

    @property
    def opt_dim(self) -> int:
        return self.action_horizon * self.action_dim

    @property
    def problem_batch_size(self) -> int:
        return int(self.config.max_batch_size)

    def get_active_js(self, full_js: JointState) -> JointState:
        return self.kinematics.get_active_js(full_js)

    def get_full_js(self, active_js: JointState) -> JointState:
        return self.kinematics.get_full_js(active_js)

    def compute_trajectory_dt(self, velocity: torch.Tensor, acceleration: torch.Tensor, jerk: torch.Tensor, dt: torch.Tensor) -> torch.Tensor:
        """the fastest time step at which a trajectory sampled at ``dt`` stays inside the velocity / acceleration / jerk limits
        (reference ``compute_trajectory_dt``; see ``TrajOptSolver.compute_trajectory_dt``)"""
        return self.solver.compute_trajectory_dt(velocity, acceleration, jerk, dt)

    def get_interpolated_trajectory(self, knots: torch.Tensor, start_position: torch.Tensor, goal_config: Optional[torch.Tensor] = None,
                                    retime: bool = True, traj_dt: Optional[torch.Tensor] = None):
        """knots [P, n_knots, dof] -> the trajectory sampled at ``interpolation_dt`` (reference ``get_interpolated_trajectory``,
        :579-634; ``TrajOptSolver.get_interpolated_trajectory``)"""
        return self.solver.get_interpolated_trajectory(knots, start_position, goal_config, retime, traj_dt)

    def reset_shape(self) -> None:
        self._solver = None

    def reset_cuda_graph(self) -> None:
        self._solver = None

    def destroy(self) -> None:
        self._solver = None

    def update_link_inertial(self, link_name: str, mass: Optional[float] = None, com=None, inertia=None) -> None:
        self.config.kinematics.kinematics_config.update_link_inertial(link_name, mass, com, inertia)

    def update_links_inertial(self, link_properties: Dict) -> None:
        self.config.kinematics.kinematics_config.update_links_inertial(link_properties)

    @property
    def solver(self) -> TrajOptSolver:
        if self._solver is None:
            c = self.config
            self._solver = TrajOptSolver.sharded(c.kinematics.kinematics_config, c.scene, c.max_batch_size, c.solver_cfg(),
                                                 use_cuda_graph=c.use_cuda_graph)
            if getattr(self, "_criteria", None):
                self._solver.update_tool_pose_criteria(self._criteria)
        return self._solver

    def _pad(self, x: Optional[torch.Tensor], batch: int) -> Optional[torch.Tensor]:
        n = self.config.max_batch_size
        if x is None or batch == n:
            return x
        return torch.cat([x, x[:1].expand(n - batch, *x.shape[1:])], dim=0)

    def _env_idx(self) -> Optional[torch.Tensor]:
        if not self.config.multi_env:
            return None
        return torch.arange(self.config.max_batch_size, device=self.config.device_cfg.device, dtype=torch.int32)

    def _wrap(self, r: TrajOptResult, batch: int, k: int, t0: float, start: torch.Tensor) -> TrajectoryOptimizerResult:
        slv, dev = self.solver, self.config.device_cfg.device
        n = self.config.max_batch_size
        D = self.action_dim
        v = lambda x, *s: x.reshape(n, k, *s)[:batch]  # noqa: E731
        H = r.position.shape[-2]
        js = JointState(v(r.position, H, D), v(r.velocity, H, D), v(r.acceleration, H, D), v(r.jerk, H, D), self.joint_names,
                        v(r.traj_dt))
        # the returned seeds re-sampled at interpolation_dt (reference interpolated_trajectory / interpolated_last_tstep)
        knots = r.knots.reshape(n * k, -1, D)
        st = start.expand(n, D).repeat_interleave(k, dim=0) if start.shape[0] != 1 else start
        goal = r.goal_config.reshape(n * k, D) if r.implicit_goal else None
        (ip, iv, ia, ij), last, _ = slv.get_interpolated_trajectory(knots, st, goal, retime=False, traj_dt=r.traj_dt.reshape(n * k))
        steps = ip.shape[1]
        w = lambda x: x.reshape(n, k, steps, D)[:batch]  # noqa: E731
        interp = JointState(w(ip), w(iv), w(ia), w(ij), self.joint_names,
                            torch.full((batch, k), self.config.interpolation_dt, device=dev))
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        total = time.perf_counter() - t0
        self.solve_time = total
        return TrajectoryOptimizerResult(
            success=v(r.success), solution=v(r.knots, self.config.n_knots, D), js_solution=js, position_error=v(r.position_error),
            rotation_error=v(r.rotation_error), interpolated_trajectory=interp, interpolated_last_tstep=last.reshape(n, k)[:batch],
            seed_cost=v(r.cost), solve_time=total, total_time=total,
            goalset_index=None if r.goalset_index is None else v(r.goalset_index),
            debug_info={"finetune_passes": r.finetune_passes, "seed_index": v(r.seed_index)})

    def solve_pose(self, goal_tool_poses: GoalToolPose, current_state: JointState, seed_config: Optional[torch.Tensor] = None,
                   seed_traj: Optional[torch.Tensor] = None, return_seeds: int = 1, num_seeds: Optional[int] = None,
                   dt: Optional[torch.Tensor] = None, use_implicit_goal: bool = False, finetune_attempts: int = 1,
                   goal_state: Optional[JointState] = None, initial_iters: Optional[int] = None,
                   time_optimal_iters: Optional[int] = None, finetune_iters: Optional[int] = None,
                   finetune_dt_scale: float = 0.55) -> TrajectoryOptimizerResult:
        """reference ``TrajOptSolver.solve_pose`` (:679-829).  ``goal_tool_poses`` [batch, 1, T, g <= max_goalset, 3 | 4];
        ``current_state.position`` [batch, dof]; ``seed_config`` [batch, n >= num_seeds, dof]; ``seed_traj`` [batch, n,
        n_knots, dof].  Without seeds and with ``use_implicit_goal=False`` the reference optimises from constant seeds at
        the current position; so does this."""
        t0 = time.perf_counter()
        if num_seeds is not None and num_seeds != self.config.num_seeds:
            raise ValueError(f"num_seeds is fixed at construction ({self.config.num_seeds}); got {num_seeds}")
        gp, gq = goal_tool_poses.static_goals(self.tool_frames)  # [batch, T, g, 3 | 4], the robot's frame order
        batch = int(gp.shape[0])
        if goal_tool_poses.num_goalset > self.config.max_goalset:
            raise ValueError(f"solve_pose: goal set of {goal_tool_poses.num_goalset} poses exceeds config.max_goalset="
                             f"{self.config.max_goalset}")
        if batch > self.config.max_batch_size:
            raise ValueError(f"solve_pose: batch_size={batch} exceeds config.max_batch_size={self.config.max_batch_size}.")
        dev = self.config.device_cfg.device
        start = current_state.position.to(dev, torch.float32).reshape(batch, -1)
        pad = lambda x: self._pad(x, batch)  # noqa: E731
        if seed_config is None and seed_traj is None:  # constant seeds at the current position (solver_core.py:206-210)
            seed_config = start.view(batch, 1, -1).expand(batch, self.config.num_seeds, start.shape[-1])
            if use_implicit_goal and goal_state is None:
                raise ValueError("use_implicit_goal needs seed_config, seed_traj or goal_state")
        r = self.solver.solve_pose(
            pad(start), pad(gp.to(dev)), pad(gq.to(dev)), env_idx=self._env_idx(),
            seed_config=pad(seed_config), seed_traj=pad(seed_traj), return_seeds=return_seeds, dt=pad(dt),
            use_implicit_goal=use_implicit_goal, finetune_attempts=finetune_attempts,
            goal_state=pad(goal_state.position.reshape(batch, -1)) if goal_state is not None else None, initial_iters=initial_iters,
            time_optimal_iters=time_optimal_iters, finetune_iters=finetune_iters, finetune_dt_scale=finetune_dt_scale)
        return self._wrap(r, batch, return_seeds, t0, pad(start))

    def solve_cspace(self, goal_state: JointState, current_state: JointState, seed_traj: Optional[torch.Tensor] = None,
                     return_seeds: int = 1, num_seeds: Optional[int] = None, dt: Optional[torch.Tensor] = None,
                     finetune_attempts: int = 1, initial_iters: Optional[int] = None, time_optimal_iters: Optional[int] = None,
                     finetune_iters: Optional[int] = None, finetune_dt_scale: float = 0.55) -> TrajectoryOptimizerResult:
        """reference ``TrajOptSolver.solve_cspace`` (:831-971)"""
        t0 = time.perf_counter()
        dev = self.config.device_cfg.device
        batch = int(current_state.position.reshape(-1, self.action_dim).shape[0])
        if batch > self.config.max_batch_size:
            raise ValueError(f"solve_cspace: batch_size={batch} exceeds config.max_batch_size={self.config.max_batch_size}.")
        start = current_state.position.to(dev, torch.float32).reshape(batch, -1)
        goal = goal_state.position.to(dev, torch.float32).reshape(batch, -1)
        pad = lambda x: self._pad(x, batch)  # noqa: E731
        r = self.solver.solve_cspace(pad(start), pad(goal), env_idx=self._env_idx(), seed_traj=pad(seed_traj),
                                     return_seeds=return_seeds, dt=pad(dt), finetune_attempts=finetune_attempts,
                                     initial_iters=initial_iters, time_optimal_iters=time_optimal_iters,
                                     finetune_iters=finetune_iters, finetune_dt_scale=finetune_dt_scale)
        return self._wrap(r, batch, return_seeds, t0, pad(start))


# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class MotionPlannerCfg:
    """reference MotionPlannerCfg (motion/motion_planner_cfg.py:28-261): the IK and trajectory-optimisation
    configurations of one robot in one (set of) world(s); no graph planner here"""

    trajopt_solver_config: TrajectoryOptimizerCfg = None
    num_ik_seeds: int = 32
    device_cfg: DeviceCfg = field(default_factory=DeviceCfg)

    @staticmethod
    def create(robot: Union[str, Dict, KinematicsCfg], scene_model: Union[str, Dict, List, None] = None,
               self_collision_check: bool = True, device_cfg: Optional[DeviceCfg] = None, num_ik_seeds: int = 32,
               num_trajopt_seeds: int = 4, position_tolerance: float = 0.005, orientation_tolerance: float = 0.05,
               use_cuda_graph: bool = True, random_seed: int = 123, optimizer_collision_activation_distance: float = 0.01,
               max_batch_size: int = 1, multi_env: bool = False, max_goalset: int = 1, assets_root: str = "", **unused
               ) -> "MotionPlannerCfg":
        """Arguments of the reference's ``MotionPlannerCfg.create`` (:37-66); task / graph-planner yaml arguments are
        accepted and ignored.  ``multi_env``: ``scene_model`` is a list of ``max_batch_size`` worlds, problem p of a batch
        plans in world p."""
        device_cfg = device_cfg or DeviceCfg()
        to = TrajectoryOptimizerCfg.create(
            robot, scene_model, num_seeds=num_trajopt_seeds, position_tolerance=position_tolerance,
            orientation_tolerance=orientation_tolerance, use_cuda_graph=use_cuda_graph, self_collision_check=self_collision_check,
            optimizer_collision_activation_distance=optimizer_collision_activation_distance, device_cfg=device_cfg,
            max_batch_size=max_batch_size, multi_env=multi_env, random_seed=random_seed, num_ik_seeds=num_ik_seeds,
            assets_root=assets_root, max_goalset=max_goalset, **{k: v for k, v in unused.items() if k in ("n_knots", "interpolation_steps", "interpolation_dt",
                                                                                  "minimum_trajectory_dt", "maximum_trajectory_dt",
                                                                                  "collision_cache")})
        if multi_env and (to.scene is None or to.scene.num_envs < max_batch_size):
            raise ValueError(f"multi_env needs a list of {max_batch_size} scene models (one world per problem)")
        return MotionPlannerCfg(trajopt_solver_config=to, num_ik_seeds=num_ik_seeds, device_cfg=device_cfg)


class _PlannerBase(ToolPoseTrackingMixin):
    def __init__(self, config: MotionPlannerCfg):
        self.config = config
        self.device_cfg = config.device_cfg
        self.trajopt_solver = TrajectoryOptimizer(config.trajopt_solver_config)
        self._ik: Optional[IKSolver] = None

    @property
    def batch_size(self) -> int:
        return self.config.trajopt_solver_config.max_batch_size

    @property
    def ik_solver(self) -> IKSolver:
        if self._ik is None:
            c = self.config.trajopt_solver_config
            self._ik = IKSolver.sharded(c.kinematics.kinematics_config, c.scene, c.max_batch_size, c.solver_cfg().ik,
                                        use_cuda_graph=c.use_cuda_graph)
            if getattr(self, "_criteria", None):
                self._ik.update_tool_pose_criteria(self._criteria)
        return self._ik

    # ---- reference properties (motion_planner.py:107-130)
    @property
    def joint_names(self) -> List[str]:
        return self.trajopt_solver.joint_names

    @property
    def action_dim(self) -> int:
        return self.trajopt_solver.action_dim

    @property
    def tool_frames(self) -> List[str]:
        return self.trajopt_solver.tool_frames

    @property
    def default_joint_state(self) -> JointState:
        return self.trajopt_solver.default_joint_state

    @property
    def kinematics(self) -> Kinematics:
        return self.trajopt_solver.kinematics

    def compute_kinematics(self, state: Union[JointState, torch.Tensor]) -> KinematicsState:
        return self.trajopt_solver.compute_kinematics(state)

    def update_world(self, scene) -> None:
        """reference ``update_world(scene_cfg)`` (:605-608): a ``SceneData`` or a scene description"""
        self.trajopt_solver.update_world(scene)
        self._ik = None

    def reset_seed(self) -> None:
        self.trajopt_solver.reset_seed()
        if self._ik is not None:
            self._ik.reset_seed()

    def clear_scene_cache(self) -> None:
        """reference ``clear_scene_cache`` (:606-609): the obstacle buffers are emptied; here the solvers are rebuilt
        without a scene on their next use"""
        self.update_world(None)

    # ---- robot model edits (reference :590-640); all in place on tensors the captured graphs read
    def enable_link_collision(self, enable_collision_links: List[str]) -> None:
        for name in enable_collision_links:
            self.kinematics.config.kinematics_config.enable_link_spheres(name)

    def disable_link_collision(self, disable_collision_links: List[str]) -> None:
        for name in disable_collision_links:
            self.kinematics.config.kinematics_config.disable_link_spheres(name)

    def update_link_inertial(self, link_name: str, mass: Optional[float] = None, com=None, inertia=None) -> None:
        self.kinematics.config.kinematics_config.update_link_inertial(link_name, mass, com, inertia)

    def update_links_inertial(self, link_properties: Dict) -> None:
        self.kinematics.config.kinematics_config.update_links_inertial(link_properties)

    def update_tool_pose_criteria(self, tool_pose_criteria: Dict[str, ToolPoseCriteria]) -> None:
        """reference ``update_tool_pose_criteria`` (:638-640): IK and trajectory optimisation score the tool frames with
        these per-axis factors from the next call on"""
        self._criteria = dict(tool_pose_criteria)
        if self._ik is not None:
            self._ik.update_tool_pose_criteria(tool_pose_criteria)
        self.trajopt_solver.update_tool_pose_criteria(tool_pose_criteria)

    def sample_configs(self, num_samples: int, rejection_ratio: int = 10) -> torch.Tensor:
        return self.trajopt_solver.sample_configs(num_samples, rejection_ratio)

    def destroy(self) -> None:
        self._ik = None
        self.trajopt_solver._solver = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()
        return False

    def _ik_seed_configs(self, goal_tool_poses: GoalToolPose, batch: int, current_state: Optional[JointState] = None):
        """IK with ``return_seeds = num_trajopt_seeds`` (L-BFGS stage always on: motion_planner.py:143-144) ->
        (success [batch, k], solution [batch, k, dof]); the batch is padded to ``max_batch_size`` with its first problem,
        a goal set to ``max_goalset`` with its last pose"""
        c = self.config.trajopt_solver_config
        n, k, dev, G = c.max_batch_size, c.num_seeds, self.device_cfg.device, c.max_goalset
        # every tool frame's goal goes to the IK stage (reference motion_planner.py:249: the whole GoalToolPose), in the
        # robot's frame order: [batch, T, g, 3 | 4]
        gp, gq = goal_tool_poses.static_goals(self.tool_frames)
        gp, gq = gp.to(dev, torch.float32), gq.to(dev, torch.float32)
        T, g = int(gp.shape[1]), int(gp.shape[2])
        if g > G:
            raise ValueError(f"goal set of {g} poses exceeds max_goalset={G}")
        if g < G:
            gp = torch.cat([gp, gp[:, :, -1:].expand(batch, T, G - g, 3)], 2)
            gq = torch.cat([gq, gq[:, :, -1:].expand(batch, T, G - g, 4)], 2)
        pad = lambda x: x if batch == n else torch.cat([x, x[:1].expand(n - batch, *x.shape[1:])], 0)  # noqa: E731
        env = torch.arange(n, device=dev, dtype=torch.int32) if c.multi_env else None
        # (the configured exit_early: reference motion_planner.py:249-253.  The robot's configuration goes with the goal as there
        #  -- ``current_state=current_state`` --: it is the first seed of the LM stage, whose ranking then prefers solutions near it.
        #  On arms whose joints turn more than a revolution (UR10e: +-2 pi) this is what keeps the goal configuration on the
        #  robot's side of the revolution: without it a plan to the pose the robot is already AT travels up to 11 rad
        #  (tools/r05/self_metric_diag.py); on the Franka it changes nothing measurable (tools/r05/planner_benchmark.py))
        cur = None
        if current_state is not None and os.environ.get("CUROBO_PLANNER_IK_CURRENT", "1") != "0":
            cur = pad(current_state.position.to(dev, torch.float32).reshape(batch, -1))
        r = self.ik_solver.solve_pose(pad(gp), pad(gq), return_seeds=k, env_idx=env, current_position=cur)
        return r.success.reshape(n, k)[:batch], r.solution.reshape(n, k, -1)[:batch]


class MotionPlanner(_PlannerBase):
    """single-problem planner with retries (reference ``MotionPlanner``, motion/motion_planner.py:38-396)"""

    def __init__(self, config: MotionPlannerCfg):
        if config.trajopt_solver_config.max_batch_size != 1:
            raise ValueError("MotionPlanner plans one problem at a time (max_batch_size must be 1); use BatchMotionPlanner")
        super().__init__(config)

    def warmup(self, enable_graph: bool = True, warmup_joint_index: int = 0, warmup_joint_delta: float = 0.2,
               num_warmup_iterations: int = 2) -> bool:
        """a few plans to a pose near the default configuration: builds the solvers and captures their graphs"""
        for _ in range(num_warmup_iterations):
            cur = JointState.from_position(self.default_joint_state.position.view(1, -1).clone(), self.joint_names)
            goal = cur.clone()
            goal.position[..., warmup_joint_index] += warmup_joint_delta
            self.plan_pose(self.compute_kinematics(goal).tool_poses.as_goal(), cur, max_attempts=1)
        return True

    def plan_pose(self, goal_tool_poses: GoalToolPose, current_state: JointState, use_implicit_goal: bool = True,
                  max_attempts: int = 5, enable_graph_attempt: int = 1) -> Optional[TrajectoryOptimizerResult]:
        """reference ``plan_pose`` / ``_plan_pose_single`` (:207-296).  Returns None when IK never found a solution."""
        if current_state.position.ndim > 2:
            raise ValueError(f"current_state must be a 2D tensor, got shape: {tuple(current_state.position.shape)}")
        t0 = time.perf_counter()
        result = None
        solve_time = 0.0
        for _ in range(max_attempts):
            ok, seed_config = self._ik_seed_configs(goal_tool_poses, 1, current_state)
            if int(ok.sum()) == 0:
                continue
            if int(ok.sum()) < ok.shape[1]:  # failed solutions are replaced by the first good one (:265-267)
                good = seed_config[ok][0:1]
                seed_config = torch.where(ok.unsqueeze(-1), seed_config, good.view(1, 1, -1))
            result = self.trajopt_solver.solve_pose(goal_tool_poses, current_state, seed_config=seed_config,
                                                    use_implicit_goal=True, finetune_attempts=1, finetune_dt_scale=0.55)
            solve_time += result.solve_time
            if int(result.success.sum()) > 0:
                break
        if result is not None:
            result.solve_time, result.total_time = solve_time, time.perf_counter() - t0
        return result

    def plan_grasp(self, grasp_poses: GoalToolPose, current_state: JointState, grasp_approach_axis: str = "z",
                   grasp_approach_offset: float = -0.15, grasp_approach_in_tool_frame: bool = True, grasp_lift_axis: str = "z",
                   grasp_lift_offset: float = -0.15, grasp_lift_in_tool_frame: bool = True, plan_approach_to_grasp: bool = True,
                   plan_grasp_to_lift: bool = True, disable_collision_links: Optional[List[str]] = None) -> GraspPlanResult:
        """reference ``plan_grasp`` (:419-588).  ``grasp_poses`` holds ``num_goalset <= max_goalset`` candidate grasps per
        tool frame.  (1) plan to the candidate set with the grasp-contact links' spheres off: finds a reachable grasp;
        (2) plan from the current state to the approach pose, the chosen grasp shifted by ``grasp_approach_offset`` along
        ``grasp_approach_axis`` (of the tool frame, or of the world); (3) from the end of that plan to the grasp under the
        linear-motion criteria (every point held on the approach line and at the grasp orientation), contact spheres off;
        (4) from the grasp to the lift pose the same way."""
        kp = self.kinematics.config.kinematics_config
        if disable_collision_links is None:
            disable_collision_links = kp.grasp_contact_link_names or []
        disable_collision_links = [n for n in disable_collision_links if n in (kp.link_names or [])]
        dev = current_state.position.device
        f = lambda: torch.tensor([False], device=dev)  # noqa: E731
        result = GraspPlanResult(success=f(), approach_success=f(), grasp_success=f(), lift_success=f())
        frames = list(grasp_poses.tool_frames)
        standard = {k: ToolPoseCriteria() for k in frames}

        def last_state(r: TrajectoryOptimizerResult) -> JointState:
            return JointState.from_position(r.js_solution.position[:, 0, -1].reshape(1, -1).clone(), joint_names=self.joint_names)

        def leg(goal: GoalToolPose, start: JointState, criteria=None, contacts_off: bool = False):
            if criteria is not None:
                self.update_tool_pose_criteria({k: criteria for k in frames})
            if contacts_off:
                self.disable_link_collision(disable_collision_links)
            try:
                return self.plan_pose(goal, start)
            finally:
                if contacts_off:
                    self.enable_link_collision(disable_collision_links)
                if criteria is not None:
                    self.update_tool_pose_criteria(standard)

        # 1: one of the grasp poses
        goalset_result = leg(grasp_poses, current_state, contacts_off=True)
        if goalset_result is None:
            result.status = "Goalset planning returned None."
            return result
        result.success = torch.zeros_like(goalset_result.success)
        result.goalset_result = goalset_result
        if not bool(goalset_result.success.any()):
            result.status = "No grasp in goal set was reachable."
            return result
        result.goalset_index = goalset_result.goalset_index.clone()
        goal_index = int(goalset_result.goalset_index.view(-1)[0].item())
        grasp = {fr: Pose(grasp_poses.position[:, 0, i, goal_index, :], grasp_poses.quaternion[:, 0, i, goal_index, :])
                 for i, fr in enumerate(frames)}

        def shifted(axis: str, offset: float, in_tool_frame: bool) -> GoalToolPose:
            off = _axis_offset_pose(axis, offset).to(dev)
            d = {fr: (p.to(dev).multiply(off) if in_tool_frame else off.multiply(p.to(dev))) for fr, p in grasp.items()}
            return GoalToolPose.from_poses(d, ordered_tool_frames=frames, num_goalset=1)

        # 2: the approach pose
        approach = leg(shifted(grasp_approach_axis, grasp_approach_offset, grasp_approach_in_tool_frame), current_state)
        result.approach_result = approach
        if approach is None or not bool(approach.success.any()):
            result.status = "Planning to approach pose failed."
            return result
        result.approach_success = approach.success.clone()
        result.approach_trajectory, result.approach_trajectory_dt = approach.js_solution, approach.js_solution.dt
        result.approach_interpolated_trajectory = approach.interpolated_trajectory
        result.approach_interpolated_last_tstep = approach.interpolated_last_tstep
        result.status = "Planning to approach pose succeeded."
        if not plan_approach_to_grasp:
            result.success = torch.ones_like(result.success)
            return result

        # 3: straight line from the approach pose to the grasp
        grasp_goal = GoalToolPose.from_poses({fr: p.to(dev) for fr, p in grasp.items()}, ordered_tool_frames=frames, num_goalset=1)
        line = ToolPoseCriteria.linear_motion(axis=grasp_approach_axis, non_terminal_scale=1.0,
                                              project_distance_to_goal=grasp_approach_in_tool_frame)
        grasp_result = leg(grasp_goal, last_state(approach), criteria=line, contacts_off=True)
        if grasp_result is None or not bool(grasp_result.success.any()):
            result.status = "Planning to grasp pose failed."
            return result
        result.grasp_trajectory, result.grasp_trajectory_dt = grasp_result.js_solution, grasp_result.js_solution.dt
        result.grasp_interpolated_trajectory = grasp_result.interpolated_trajectory
        result.grasp_interpolated_last_tstep = grasp_result.interpolated_last_tstep
        result.success = grasp_result.success.clone()
        result.grasp_success = grasp_result.success.clone()
        result.status = "Planning to grasp pose succeeded."
        if not plan_grasp_to_lift:
            return result

        # 4: straight line from the grasp to the lift pose
        lift_line = ToolPoseCriteria.linear_motion(axis=grasp_lift_axis, non_terminal_scale=1.0,
                                                   project_distance_to_goal=grasp_lift_in_tool_frame)
        lift = leg(shifted(grasp_lift_axis, grasp_lift_offset, grasp_lift_in_tool_frame), last_state(grasp_result),
                   criteria=lift_line, contacts_off=True)
        if lift is None or not bool(lift.success.any()):
            result.status = "Planning to lift pose failed."
            result.success = torch.zeros_like(result.success)
            return result
        result.lift_trajectory, result.lift_trajectory_dt = lift.js_solution, lift.js_solution.dt
        result.lift_interpolated_trajectory = lift.interpolated_trajectory
        result.lift_interpolated_last_tstep = lift.interpolated_last_tstep
        result.success = lift.success.clone()
        result.lift_success = lift.success.clone()
        result.status = "Planning to lift pose succeeded."
        return result

    def plan_cspace(self, goal_state: JointState, current_state: JointState, max_attempts: int = 5,
                    enable_graph_attempt: int = 1) -> Optional[TrajectoryOptimizerResult]:
        """reference ``plan_cspace`` (:329-396): three finetune passes at a dt scale of 0.75"""
        if current_state.position.ndim > 2 or goal_state.position.ndim > 2:
            raise ValueError("current_state and goal_state must be 2D tensors")
        t0 = time.perf_counter()
        result, solve_time = None, 0.0
        for _ in range(max_attempts):
            result = self.trajopt_solver.solve_cspace(goal_state, current_state, finetune_attempts=3, finetune_dt_scale=0.75)
            solve_time += result.solve_time
            if int(result.success.sum()) > 0:
                break
        if result is not None:
            result.solve_time, result.total_time = solve_time, time.perf_counter() - t0
        return result


class BatchMotionPlanner(_PlannerBase):
    """``max_batch_size`` independent problems per pass (reference ``BatchMotionPlanner``, motion_planner_batch.py:38-289)"""

    def warmup(self, enable_graph: bool = True, num_warmup_iterations: int = 2) -> bool:
        n = self.batch_size
        for _ in range(num_warmup_iterations):
            cur = JointState.from_position(self.default_joint_state.position.view(1, -1).repeat(n, 1), self.joint_names)
            goal = cur.clone()
            goal.position[..., 0] += 0.2
            self.plan_cspace(goal, cur)
        return True

    def plan_pose(self, goal_tool_poses: GoalToolPose, current_state: JointState, use_implicit_goal: bool = True,
                  max_attempts: int = 1, success_ratio: float = 1.0, enable_graph_attempt: int = 0
                  ) -> Optional[TrajectoryOptimizerResult]:
        """reference ``plan_pose`` (:139-221): up to ``max_attempts`` IK -> trajopt passes over the whole batch, a
        problem keeps the result of the first pass that solved it; stops when ``success_ratio`` of the batch is solved.
        Returns None when IK never found a solution."""
        t0 = time.perf_counter()
        batch = goal_tool_poses.batch_size
        best: Optional[TrajectoryOptimizerResult] = None
        solved = torch.zeros(batch, dtype=torch.bool, device=self.device_cfg.device)
        for _ in range(max_attempts):
            ok, seed_config = self._ik_seed_configs(goal_tool_poses, batch, current_state)
            if int(ok.sum()) == 0:
                continue
            # per problem, failed IK solutions are replaced by that problem's first good one (reference :265-267, per row)
            first_good = torch.argmax(ok.to(torch.int8), dim=1)
            good = seed_config[torch.arange(batch, device=seed_config.device), first_good]
            seed_config = torch.where((ok | ~ok.any(dim=1, keepdim=True)).unsqueeze(-1), seed_config, good.unsqueeze(1))
            r = self.trajopt_solver.solve_pose(goal_tool_poses, current_state, seed_config=seed_config,
                                               use_implicit_goal=use_implicit_goal)
            if best is None:
                best, solved = r, r.success.any(dim=-1)
            else:
                newly = r.success.any(dim=-1) & ~solved
                if bool(newly.any()):
                    best.copy_at_batch_indices(r, newly)
                    solved = solved | newly
            if float(solved.float().mean()) >= success_ratio:
                break
        if best is not None:
            best.total_time = time.perf_counter() - t0
        return best

    def plan_grasp(self, grasp_poses: GoalToolPose, current_state: JointState, grasp_approach_axis: str = "z",
                   grasp_approach_offset: float = -0.15, grasp_approach_in_tool_frame: bool = True, grasp_lift_axis: str = "z",
                   grasp_lift_offset: float = -0.15, grasp_lift_in_tool_frame: bool = True, plan_approach_to_grasp: bool = True,
                   plan_grasp_to_lift: bool = True, disable_collision_links: Optional[List[str]] = None) -> GraspPlanResult:
        """reference batch ``plan_grasp`` (motion_planner_batch.py:291-472): the four stages of ``MotionPlanner.plan_grasp`` for
        every problem of the batch at once.  Every problem is planned at every stage (fixed shapes); a problem that failed a
        stage gets the pose it is already in as the goal of the next ones, and its success flags stay off."""
        kp = self.kinematics.config.kinematics_config
        if disable_collision_links is None:
            disable_collision_links = kp.grasp_contact_link_names or []
        disable_collision_links = [n for n in disable_collision_links if n in (kp.link_names or [])]
        batch, dev = grasp_poses.batch_size, current_state.position.device
        frames = list(grasp_poses.tool_frames)
        f = lambda: torch.zeros(batch, dtype=torch.bool, device=dev)  # noqa: E731
        result = GraspPlanResult(success=f(), approach_success=f(), grasp_success=f(), lift_success=f())
        standard = {k: ToolPoseCriteria() for k in frames}

        def leg(goal: GoalToolPose, start: JointState, criteria=None, contacts_off: bool = False):
            if criteria is not None:
                self.update_tool_pose_criteria({k: criteria for k in frames})
            if contacts_off:
                self.disable_link_collision(disable_collision_links)
            try:
                return self.plan_pose(goal, start)
            finally:
                if contacts_off:
                    self.enable_link_collision(disable_collision_links)
                if criteria is not None:
                    self.update_tool_pose_criteria(standard)

        def last_state(r: TrajectoryOptimizerResult) -> JointState:
            return JointState.from_position(r.js_solution.position[:, 0, -1].clone(), joint_names=self.joint_names)

        def stay_where_failed(goal: GoalToolPose, state: JointState, failed: torch.Tensor) -> None:
            if not bool(failed.any()):
                return
            here = self.compute_kinematics(state).tool_poses  # [batch, 1, T, 3 | 4]
            for i, fr in enumerate(goal.tool_frames):
                j = list(here.tool_frames).index(fr)
                goal.position[failed, :, i, :, :] = here.position[failed, 0, j][:, None, None, :]
                goal.quaternion[failed, :, i, :, :] = here.quaternion[failed, 0, j][:, None, None, :]

        # 1: a reachable grasp per problem
        goalset_result = leg(grasp_poses, current_state, contacts_off=True)
        if goalset_result is None:
            result.status = "Goalset planning returned None."
            return result
        goalset_ok = goalset_result.success.any(dim=-1)
        result.goalset_result = goalset_result
        if not bool(goalset_ok.any()):
            result.status = "No grasp in goal set was reachable."
            return result
        result.goalset_index = goalset_result.goalset_index.clone()
        idx = goalset_result.goalset_index[:, 0].long().clone()
        idx[~goalset_ok] = 0
        rows = torch.arange(batch, device=dev)
        grasp = {fr: Pose(grasp_poses.position[:, 0, i].to(dev)[rows, idx], grasp_poses.quaternion[:, 0, i].to(dev)[rows, idx])
                 for i, fr in enumerate(frames)}

        def shifted(axis: str, offset: float, in_tool_frame: bool) -> GoalToolPose:
            off = _axis_offset_pose(axis, offset).to(dev)
            d = {fr: (p.multiply(off) if in_tool_frame else off.multiply(p)) for fr, p in grasp.items()}
            return GoalToolPose.from_poses(d, ordered_tool_frames=frames, num_goalset=1)

        # 2: the approach poses
        approach = leg(shifted(grasp_approach_axis, grasp_approach_offset, grasp_approach_in_tool_frame), current_state)
        result.approach_result = approach
        if approach is None:
            result.status = "Planning to approach pose failed."
            return result
        approach_ok = goalset_ok & approach.success.any(dim=-1)
        result.approach_success = approach_ok.clone()
        result.approach_trajectory, result.approach_trajectory_dt = approach.js_solution, approach.js_solution.dt
        result.approach_interpolated_trajectory = approach.interpolated_trajectory
        result.approach_interpolated_last_tstep = approach.interpolated_last_tstep
        if not plan_approach_to_grasp:
            result.success = approach_ok.clone()
            result.status = "Planning to approach pose completed."
            return result

        # 3: straight lines from the approach poses to the grasps
        approach_end = last_state(approach)
        grasp_goal = GoalToolPose.from_poses({fr: p.clone() for fr, p in grasp.items()}, ordered_tool_frames=frames, num_goalset=1)
        stay_where_failed(grasp_goal, approach_end, ~approach_ok)
        line = ToolPoseCriteria.linear_motion(axis=grasp_approach_axis, non_terminal_scale=1.0,
                                              project_distance_to_goal=grasp_approach_in_tool_frame)
        grasp_result = leg(grasp_goal, approach_end, criteria=line, contacts_off=True)
        if grasp_result is None:
            result.status = "Planning to grasp pose failed."
            return result
        grasp_ok = approach_ok & grasp_result.success.any(dim=-1)
        result.grasp_success = grasp_ok.clone()
        result.grasp_trajectory, result.grasp_trajectory_dt = grasp_result.js_solution, grasp_result.js_solution.dt
        result.grasp_interpolated_trajectory = grasp_result.interpolated_trajectory
        result.grasp_interpolated_last_tstep = grasp_result.interpolated_last_tstep
        if not plan_grasp_to_lift:
            result.success = grasp_ok.clone()
            result.status = "Planning to grasp pose completed."
            return result

        # 4: lift
        lift_start = last_state(grasp_result)
        lift_goal = shifted(grasp_lift_axis, grasp_lift_offset, grasp_lift_in_tool_frame)
        stay_where_failed(lift_goal, lift_start, ~grasp_ok)
        lift_line = ToolPoseCriteria.linear_motion(axis=grasp_lift_axis, non_terminal_scale=1.0,
                                                   project_distance_to_goal=grasp_lift_in_tool_frame)
        lift = leg(lift_goal, lift_start, criteria=lift_line, contacts_off=True)
        if lift is None:
            result.status = "Planning to lift pose failed."
            return result
        lift_ok = grasp_ok & lift.success.any(dim=-1)
        result.lift_success = lift_ok.clone()
        result.lift_trajectory, result.lift_trajectory_dt = lift.js_solution, lift.js_solution.dt
        result.lift_interpolated_trajectory = lift.interpolated_trajectory
        result.lift_interpolated_last_tstep = lift.interpolated_last_tstep
        result.success = lift_ok.clone()
        result.status = "Grasp planning completed."
        return result

    def plan_cspace(self, goal_states: JointState, current_state: JointState, max_attempts: int = 1,
                    success_ratio: float = 1.0, enable_graph_attempt: int = 0) -> Optional[TrajectoryOptimizerResult]:
        """reference ``plan_cspace`` (:223-289)"""
        t0 = time.perf_counter()
        batch = int(current_state.position.shape[0])
        best: Optional[TrajectoryOptimizerResult] = None
        solved = torch.zeros(batch, dtype=torch.bool, device=self.device_cfg.device)
        for _ in range(max_attempts):
            r = self.trajopt_solver.solve_cspace(goal_states, current_state)
            if best is None:
                best, solved = r, r.success.any(dim=-1)
            else:
                newly = r.success.any(dim=-1) & ~solved
                if bool(newly.any()):
                    best.copy_at_batch_indices(r, newly)
                    solved = solved | newly
            if float(solved.float().mean()) >= success_ratio:
                break
        if best is not None:
            best.total_time = time.perf_counter() - t0
        return best
