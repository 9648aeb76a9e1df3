"""``ModelPredictiveControl`` / ``ModelPredictiveControlCfg``: the reference's MPC front end over ``solver.mpc.MPCSolver``.

Reference: ``curobo/model_predictive_control.py`` (``MPCSolver`` / ``MPCSolverCfg.create``, ``curobo/_src/solver/solver_mpc.py:33-878``,
``solver_mpc_cfg.py:126-287``):

    config = ModelPredictiveControlCfg.create(robot="franka.yml", scene_model=..., optimization_dt=0.02, interpolation_steps=4)
    mpc = ModelPredictiveControl(config)
    mpc.setup(current_state)
    mpc.update_goal_tool_poses(goal_tool_poses)
    result = mpc.optimize_next_action(current_state)      # or optimize_action_sequence

``update_goal_tool_poses`` has the reference's defaults (``run_ik=True, use_ik_goal=True``): a collision-free IK solution of the
goal, close to the current configuration, becomes the goal configuration of the joint-position tracking term beside the pose
tracking; ``run_ik=False`` tracks the poses alone.  ``update_goal_state`` + ``enable_joint_position_tracking`` (+
``disable_tool_pose_tracking``) give joint-space control."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

import torch

from .kinematics import Kinematics, KinematicsCfg, KinematicsState
from .motion_planner import _load_kinematics, apply_robot_limits
from .scene import SceneData
from .scene.config import scene_from_config
from .solver.mpc import MPCSolver, MPCSolverCfg, MPCSolverResult
from .types import DeviceCfg, GoalToolPose, JointState, Pose, ToolPoseCriteria
from .solver.tracking import ToolPoseTrackingMixin

ModelPredictiveControlResult = MPCSolverResult


@dataclass
class ModelPredictiveControlCfg:
    kinematics: KinematicsCfg = None
    scene: Optional[SceneData] = None
    device_cfg: DeviceCfg = field(default_factory=DeviceCfg)
    max_batch_size: int = 1
    solver: MPCSolverCfg = field(default_factory=MPCSolverCfg)

    @staticmethod
    def create(robot: Union[str, Dict, KinematicsCfg], scene_model: Union[str, Dict, List, None] = None,
               self_collision_check: bool = True, device_cfg: Optional[DeviceCfg] = None, use_cuda_graph: bool = True,
               optimizer_collision_activation_distance: float = 0.01, optimization_dt: float = 0.02, interpolation_steps: int = 4,
               num_control_points: Optional[int] = None, warm_start_optimization_num_iters: Optional[int] = None,
               cold_start_optimization_num_iters: Optional[int] = None, max_batch_size: int = 1, max_goalset: int = 1,
               assets_root: str = "", continuous_commands: bool = False, task: str = "package", **unused) -> "ModelPredictiveControlCfg":
        """Arguments of the reference's ``MPCSolverCfg.create`` (solver_mpc_cfg.py:126-165).  Task / optimiser yaml arguments are
        accepted and ignored (the cost set and optimiser of ``content/configs/task/mpc/`` are built in); the iteration counts
        default to this backend's (100 cold / 25 warm L-BFGS iterations: its line search evaluates four step sizes per
        iteration).  ``continuous_commands``: see ``MPCSolverCfg`` (False = the reference's command indexing).
        ``task``: "package" = this package's MPC task values, "reference" = ``MPCSolverCfg.reference_task()``."""
        if max_goalset != 1:
            raise ValueError("the MPC front end tracks one goal pose per tool frame (max_goalset must be 1)")
        device_cfg = device_cfg or DeviceCfg()
        kin = _load_kinematics(robot, device_cfg.device, assets_root)
        if task not in ("package", "reference"):
            raise ValueError(f"task must be 'package' or 'reference', got {task!r}")
        make = MPCSolverCfg.reference_task if task == "reference" else MPCSolverCfg  # (lbfgs_mpc.yml's values + 300 / 200 iterations)
        s = make(optimization_dt=optimization_dt, interpolation_steps=interpolation_steps, use_cuda_graph=use_cuda_graph,
                 continuous_commands=continuous_commands)
        if num_control_points is not None:
            s.n_knots = int(num_control_points)
        if warm_start_optimization_num_iters is not None:
            s.warm_start_optimization_num_iters = int(warm_start_optimization_num_iters)
        if cold_start_optimization_num_iters is not None:
            s.cold_start_optimization_num_iters = int(cold_start_optimization_num_iters)
        s.rollout.scene_activation_distance = optimizer_collision_activation_distance
        apply_robot_limits(s.rollout, kin)
        if not self_collision_check:
            s.rollout.self_collision_weight = 0.0
        return ModelPredictiveControlCfg(kinematics=kin, scene=scene_from_config(scene_model, device_cfg.device, cache=unused.get("collision_cache")),
                                         device_cfg=device_cfg,
                                         max_batch_size=max_batch_size, solver=s)


class ModelPredictiveControl(ToolPoseTrackingMixin):
    """the reference's ``MPCSolver`` call surface for ``max_batch_size`` robots"""

    _tracking_non_terminal_factor = 1.0  # (a controller tracks the pose along the whole horizon: solver/mpc.py's non_terminal_pose_factor)


    def __init__(self, config: ModelPredictiveControlCfg):
        self.config = config
        self.kinematics = Kinematics(config.kinematics, compute_spheres=True)
        self._solver: Optional[MPCSolver] = None
        self._goal: Optional[GoalToolPose] = None
        self._criteria: Optional[Dict[str, ToolPoseCriteria]] = None

    @property
    def solver(self) -> MPCSolver:
        if self._solver is None:
            c = self.config
            self._solver = MPCSolver(c.kinematics.kinematics_config, c.scene, c.max_batch_size, c.solver)
            if self._criteria:
                self._apply_criteria()
        return self._solver

    # ---- reference properties
    @property
    def device_cfg(self) -> DeviceCfg:
        return self.config.device_cfg

    @property
    def joint_names(self) -> List[str]:
        return self.kinematics.joint_names

    @property
    def tool_frames(self) -> List[str]:
        return self.kinematics.tool_frames

    @property
    def action_dim(self) -> int:
        return self.kinematics.dof

    @property
    def action_horizon(self) -> int:
        return self.config.solver.n_knots

    @property
    def problem_batch_size(self) -> int:
        return int(self.config.max_batch_size)

    @property
    def command_dt(self) -> float:
        return self.config.solver.optimization_dt / self.config.solver.interpolation_steps

    @property
    def default_joint_position(self) -> torch.Tensor:
        return self.kinematics.default_joint_position

    @property
    def default_joint_state(self) -> JointState:
        return self.kinematics.default_joint_state

    def compute_kinematics(self, state: Union[JointState, torch.Tensor]) -> KinematicsState:
        return self.kinematics.compute_kinematics(state)

    def get_active_js(self, full_js: JointState) -> JointState:
        return self.kinematics.get_active_js(full_js)

    def get_full_js(self, active_js: JointState) -> JointState:
        return self.kinematics.get_full_js(active_js)

    # ---- problem definition
    def _batch(self, state: JointState) -> JointState:
        n, D = self.config.max_batch_size, self.action_dim
        f = lambda t: None if t is None else t.to(self.device_cfg.device, torch.float32).reshape(-1, D).expand(n, D).contiguous()  # noqa: E731
        return JointState(f(state.position), f(state.velocity), f(state.acceleration), None, state.joint_names)

    def setup(self, current_state: JointState, goal_tool_poses: Optional[GoalToolPose] = None) -> None:
        """reference ``setup`` (:261-330).  Without a goal the robot holds the tool poses of ``current_state``"""
        state = self._batch(current_state)
        if goal_tool_poses is None:
            goal_tool_poses = self.compute_kinematics(JointState.from_position(state.position)).tool_poses.as_goal()
        self._goal = goal_tool_poses.clone()
        self.solver.setup(state, self._goal)

    def set_default_goal_from_current_state(self, current_state: JointState) -> None:
        """hold the current tool pose.  No goal IK here: the goal IS the pose of the current configuration, and an IK that fails on
        one robot must not leave every robot on its old goal without a word (``update_goal_tool_poses`` returns False then)"""
        ok = self.update_goal_tool_poses(self.compute_kinematics(JointState.from_position(self._batch(current_state).position)).tool_poses.as_goal(),
                                         run_ik=False)
        if not ok:
            raise RuntimeError("set_default_goal_from_current_state: the current tool pose was not accepted as a goal")

    def update_goal_tool_poses(self, goal_tool_poses: Union[GoalToolPose, Dict[str, Pose]], robot_ids: Optional[torch.Tensor] = None,
                               run_ik: bool = True, use_ik_goal: bool = True, use_best_effort_ik: bool = False) -> bool:
        """reference ``update_goal_tool_poses`` (:365-438): the whole goal, or with ``robot_ids`` the rows of those robots.  A
        dictionary ``{tool frame: Pose}`` is taken as one goal per robot."""
        if isinstance(goal_tool_poses, dict):
            goal_tool_poses = GoalToolPose.from_poses({k: Pose(p.position.reshape(-1, 3), p.quaternion.reshape(-1, 4)) for k, p in goal_tool_poses.items()},
                                                      ordered_tool_frames=[f for f in self.tool_frames if f in goal_tool_poses])
        if robot_ids is not None:
            if self._goal is None:
                raise ValueError("goal_tool_poses not set, call update_goal_tool_poses without robot_ids first")
            candidate = self._goal.clone()
            ids = robot_ids.to(candidate.position.device).long()
            for i, name in enumerate(goal_tool_poses.tool_frames):
                j = candidate.tool_frames.index(name)
                candidate.position[ids, :, j] = goal_tool_poses.position[ids, :, i].to(candidate.position.device)
                candidate.quaternion[ids, :, j] = goal_tool_poses.quaternion[ids, :, i].to(candidate.position.device)
        else:
            candidate = goal_tool_poses.clone()
        # (run_ik: the goal's IK solution becomes the tracked goal configuration; a failed IK leaves the previous goal in place)
        if not self.solver.update_goal_tool_poses(candidate, run_ik=run_ik, use_ik_goal=use_ik_goal, use_best_effort_ik=use_best_effort_ik):
            return False
        self._goal = candidate
        return True

    def update_goal_state(self, goal_state: JointState, robot_ids: Optional[torch.Tensor] = None) -> None:
        """goal configuration of the joint-position tracking (reference :458-474; ``robot_ids`` is not supported there either)"""
        if robot_ids is not None:
            raise ValueError("robot_ids not supported for update_goal_state")
        self.solver.update_goal_state(self._batch(goal_state))

    def update_seed_trajectory(self, seed_trajectory: torch.Tensor) -> None:
        """knots [batch, action_horizon, action_dim] the next solve starts from (reference solver_mpc.py:498-514)"""
        self.solver.update_seed_trajectory(seed_trajectory)

    def prepare_safe_deceleration_trajectory(self, current_state: JointState, failed_mask: Optional[torch.Tensor] = None,
                                             deceleration_time: Optional[float] = None, deceleration_profile: Optional[str] = None) -> torch.Tensor:
        """knots that bring the robots to rest from ``current_state`` (reference solver_mpc.py:701-762); see ``MPCSolver``"""
        return self.solver.prepare_safe_deceleration_trajectory(self._batch(current_state), failed_mask, deceleration_time, deceleration_profile)

    def update_seed_trajectory_from_goal_state(self, goal_joint_state: JointState) -> None:
        """seed the next solve with the straight joint-space line from the current state to ``goal_joint_state`` (reference :516-531)"""
        self.solver.update_seed_trajectory_from_goal_state(self._batch(goal_joint_state))

    def enable_joint_position_tracking(self) -> None:
        self.solver.enable_joint_position_tracking()

    def disable_joint_position_tracking(self) -> None:
        self.solver.disable_joint_position_tracking()

    def update_current_state(self, current_state: JointState) -> None:
        self.solver.update_current_state(self._batch(current_state))

    def reset_robot(self, current_state: JointState) -> None:
        self.solver.reset_robot(self._batch(current_state))

    def update_world(self, scene) -> None:
        self.config.scene = scene if (scene is None or isinstance(scene, SceneData)) else scene_from_config(scene, self.device_cfg.device)
        self._solver = None

    # ---- solves
    def optimize_next_action(self, current_state: JointState) -> MPCSolverResult:
        return self.solver.optimize_next_action(self._batch(current_state))

    def optimize_action_sequence(self, current_state: JointState) -> MPCSolverResult:
        return self.solver.optimize_action_sequence(self._batch(current_state))

    def cold_start_solve(self, current_state: JointState) -> None:
        self.solver.cold_start_solve(self._batch(current_state))

    def warm_start_solve(self, current_state: JointState) -> None:
        self.solver.warm_start_solve(self._batch(current_state))

    # ---- robot model / cost edits
    def _apply_criteria(self) -> None:
        for r in (self._solver.rollout, self._solver.metrics_rollout):
            r.update_tool_pose_criteria(self._criteria)

    def update_tool_pose_criteria(self, tool_pose_criteria: Dict[str, ToolPoseCriteria]) -> None:
        self._criteria = dict(tool_pose_criteria)
        if self._solver is not None:
            self._apply_criteria()

    def update_link_inertial(self, link_name: str, mass: Optional[float] = None, com=None, inertia=None) -> None:
        self.config.kinematics.kinematics_config.update_link_inertial(link_name, mass, com, inertia)

    def update_links_inertial(self, link_properties: Dict) -> None:
        self.config.kinematics.kinematics_config.update_links_inertial(link_properties)

    def reset_seed(self) -> None:
        """(the controller draws no random numbers: its seeds are the hold-still and the shifted plans)"""

    def reset_shape(self) -> None:
        self._solver = None

    def reset_cuda_graph(self) -> None:
        self._solver = None

    def destroy(self) -> None:
        self._solver = None
