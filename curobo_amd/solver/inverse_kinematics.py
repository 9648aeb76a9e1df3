"""``InverseKinematics`` / ``InverseKinematicsCfg``: the reference's IK front end (``curobo.inverse_kinematics``:
``IKSolver`` / ``IKSolverCfg.create`` / ``IKSolverResult``, reference ``curobo/_src/solver/solver_ik.py:60-760``,
``solver_ik_cfg.py:30-330``) over ``curobo_amd.solver.IKSolver``.  The call sequence of the reference's
``benchmark/ik_benchmark.py:55-142`` runs unchanged: ``create(robot=..., scene_model=..., num_seeds=...)`` ->
``sample_configs`` -> ``compute_kinematics(JointState)`` -> ``tool_poses.as_goal()`` -> ``solve_pose(goal_tool_poses=...)``."""

from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

import torch

from ..collision_checking import RobotCollisionChecker
from ..kinematics import Kinematics, KinematicsCfg, KinematicsState
from ..scene import SceneData
from ..scene.config import scene_from_config
from ..types import DeviceCfg, GoalToolPose, JointState
from .ik import IKSolver, IKSolverCfg
from .tracking import ToolPoseTrackingMixin


@dataclass
class InverseKinematicsResult:
    """reference IKSolverResult / BaseSolverResult fields (solver_ik_result.py, solver_result.py)"""

    success: torch.Tensor            # [batch, return_seeds] bool
    solution: torch.Tensor           # [batch, return_seeds, dof]
    js_solution: JointState
    position_error: torch.Tensor     # [batch, return_seeds] metres
    rotation_error: torch.Tensor     # [batch, return_seeds] radians
    goalset_index: Optional[torch.Tensor] = None
    solve_time: float = 0.0
    debug_info: Optional[dict] = None
    total_time: float = 0.0
    position_tolerance: float = 0.0
    orientation_tolerance: float = 0.0
    batch_size: int = 0
    num_seeds: int = 0

    _TENSORS = ("success", "solution", "position_error", "rotation_error", "goalset_index")

    def clone(self) -> "InverseKinematicsResult":
        """deep copy (reference BaseSolverResult.clone, solver_base_result.py:81-131)"""
        import dataclasses

        c = lambda t: None if t is None else t.clone()  # noqa: E731
        dbg = None if self.debug_info is None else {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.debug_info.items()}
        return dataclasses.replace(self, **{f: c(getattr(self, f)) for f in self._TENSORS}, js_solution=c(self.js_solution), debug_info=dbg)

    def copy_successful_solutions(self, other: "InverseKinematicsResult") -> None:
        """every (problem, seed) entry that succeeded in ``other`` replaces the entry here, in place (reference :133-210: merging the
        results of several attempts)"""
        if self.success is None or other.success is None:
            raise ValueError("success is not set")
        b, k = other.success.nonzero(as_tuple=True)
        for f in self._TENSORS:
            dst, src = getattr(self, f), getattr(other, f)
            if dst is not None and src is not None:
                dst[b, k] = src[b, k]
        if self.js_solution is not None and other.js_solution is not None:
            self.js_solution.copy_at_batch_seed_indices(other.js_solution, b, k)

    def copy_at_batch_indices(self, other: "InverseKinematicsResult", mask: torch.Tensor) -> None:
        """whole problems (all their seeds) selected by ``mask`` [batch] taken from ``other``, in place (reference :212-240:
        first-success-wins merging in batched planning)"""
        for f in self._TENSORS:
            dst, src = getattr(self, f), getattr(other, f)
            if dst is not None and src is not None and dst.shape == src.shape:
                dst[mask] = src[mask]
        if self.js_solution is not None and other.js_solution is not None:
            for f in ("position", "velocity", "acceleration", "jerk"):
                dst, src = getattr(self.js_solution, f), getattr(other.js_solution, f)
                if dst is not None and src is not None and dst.shape == src.shape:
                    dst[mask] = src[mask]

    def get_unique_solution(self, roundoff_decimals: int = 2) -> torch.Tensor:
        """the successful solutions, one representative per configuration after rounding to ``roundoff_decimals`` (reference
        ``get_unique_solution``, solver_ik_result.py) -> [num_unique, dof]"""
        sol = self.solution[self.success]
        if sol.ndim != 2:
            raise ValueError("Solution shape is not of length 2")
        rounded = torch.round(sol, decimals=roundoff_decimals)
        uniq, inverse = torch.unique(rounded, dim=-2, return_inverse=True)
        first = torch.full((uniq.shape[0],), sol.shape[0], dtype=torch.long, device=sol.device)
        first.scatter_reduce_(0, inverse, torch.arange(sol.shape[0], device=sol.device), reduce="amin")
        return sol[first]


@dataclass
class InverseKinematicsCfg:
    kinematics: KinematicsCfg = None
    scene: Optional[SceneData] = None
    device_cfg: DeviceCfg = field(default_factory=DeviceCfg)
    num_seeds: int = 32
    position_tolerance: float = 0.005
    orientation_tolerance: float = 0.05
    use_cuda_graph: bool = True
    self_collision_check: bool = True
    optimizer_collision_activation_distance: float = 0.0025
    exit_early: bool = True
    seed_solver_num_seeds: int = 64
    use_lm_seed: bool = True
    max_batch_size: int = 0
    max_goalset: int = 1
    stream_shards: int = 1
    #: reference ``override_iters_for_multi_link_ik``: L-BFGS iterations raised to this when lower
    override_iters_for_multi_link_ik: Optional[int] = None

    @staticmethod
    def create(robot: Union[str, Dict], scene_model: Union[str, Dict, List, None] = None, num_seeds: int = 32,
               position_tolerance: float = 0.005, orientation_tolerance: float = 0.05, use_cuda_graph: bool = True,
               self_collision_check: bool = True, optimizer_collision_activation_distance: float = 0.0025,
               device_cfg: Optional[DeviceCfg] = None, seed_solver_num_seeds: Optional[int] = None, max_batch_size: int = 0,
               max_goalset: int = 1, use_lm_seed: bool = True, exit_early: bool = True, assets_root: str = "",
               override_iters_for_multi_link_ik: Optional[int] = None, **unused) -> "InverseKinematicsCfg":
        """``robot``: packaged name (``"franka.yml"``), a robot yaml path or its dictionary.  ``scene_model``: the
        reference's scene format (see ``curobo_amd.scene.config``).  Keyword arguments of the reference this backend has
        no use for (``optimizer_configs``, ``metrics_rollout``, ``transition_model``, ...) are accepted and ignored: the
        cost set and optimiser settings of ``content/configs/task/ik/lbfgs_ik.yml`` are built in."""
        import os

        device_cfg = device_cfg or DeviceCfg()
        dev = device_cfg.device
        if isinstance(robot, dict):
            kin = KinematicsCfg.from_data_dict(robot, assets_root=assets_root, device=dev)
        elif os.path.exists(str(robot)):
            kin = KinematicsCfg.from_robot_yaml_file(robot, assets_root or os.path.dirname(os.path.abspath(robot)), device=dev)
        else:
            kin = KinematicsCfg.from_packaged(str(robot).replace(".yml", "").replace(".yaml", ""), device=dev)
        scene = scene_from_config(scene_model, dev, cache=unused.get("collision_cache"))
        return InverseKinematicsCfg(
            kinematics=kin, scene=scene, device_cfg=device_cfg, num_seeds=num_seeds, position_tolerance=position_tolerance,
            orientation_tolerance=orientation_tolerance, use_cuda_graph=use_cuda_graph, self_collision_check=self_collision_check,
            optimizer_collision_activation_distance=optimizer_collision_activation_distance, exit_early=exit_early,
            seed_solver_num_seeds=seed_solver_num_seeds or max(32, 2 * num_seeds), use_lm_seed=use_lm_seed,
            max_batch_size=max_batch_size, max_goalset=max_goalset, override_iters_for_multi_link_ik=override_iters_for_multi_link_ik)


class InverseKinematics(ToolPoseTrackingMixin):
    def __init__(self, config: InverseKinematicsCfg):
        self.config = config
        self.kinematics = Kinematics(config.kinematics, compute_spheres=True)
        self._checker: Optional[RobotCollisionChecker] = None
        self._solvers: Dict[int, IKSolver] = {}
        self.solve_time = 0.0

    # ---- reference members used by benchmarks / planners
    @property
    def joint_names(self) -> List[str]:
        return self.kinematics.joint_names

    @property
    def tool_frames(self) -> List[str]:
        return self.kinematics.tool_frames

    @property
    def dof(self) -> int:
        return self.config.kinematics.kinematics_config.num_dof

    def compute_kinematics(self, joint_state: Union[JointState, torch.Tensor]) -> KinematicsState:
        return self.kinematics.compute_kinematics(joint_state)

    def _collision_checker(self) -> RobotCollisionChecker:
        if self._checker is None:
            self._checker = RobotCollisionChecker(self.config.kinematics, self.config.scene)
        return self._checker

    def sample_configs(self, num_samples: int, rejection_ratio: int = 10) -> torch.Tensor:
        """collision-free joint configurations [<= num_samples, dof] (reference IKSolver.sample_configs)"""
        chk = self._collision_checker()
        chk.rejection_ratio = rejection_ratio
        return chk.sample(num_samples, mask_valid=True)

    def reset_seed(self) -> None:
        for s in self._solvers.values():
            s._gen.manual_seed(s.cfg.seed)
            s.reset_seed()

    def update_world(self, scene) -> None:
        """``SceneData``, or a scene description (``curobo.scene.Scene``, dictionary, yaml path)"""
        if scene is not None and not isinstance(scene, SceneData):
            from ..scene.config import scene_from_config

            scene = scene_from_config(scene, self.config.kinematics.kinematics_config.device)
        self.config.scene = scene
        self._solvers.clear()
        self._checker = None

    # ---- reference solver_ik.py:184-216, 253-257, 293-296
    @property
    def action_dim(self) -> int:
        return self.dof

    @property
    def action_horizon(self) -> int:
        return 1

    @property
    def default_joint_state(self) -> JointState:
        from ..workloads import start_configuration

        q = torch.as_tensor(start_configuration(self.config.kinematics.model), device=self.config.kinematics.kinematics_config.device)
        return JointState.from_position(q, joint_names=self.joint_names)

    @property
    def device_cfg(self) -> DeviceCfg:
        return self.config.device_cfg

    @property
    def default_joint_position(self) -> torch.Tensor:
        return self.kinematics.default_joint_position

    @property
    def problem_batch_size(self) -> int:
        return max(self._solvers) if self._solvers else int(self.config.max_batch_size)

    def get_active_js(self, full_js: JointState) -> JointState:
        return self.kinematics.get_active_js(full_js)

    def get_full_js(self, active_js: JointState) -> JointState:
        return self.kinematics.get_full_js(active_js)

    def reset_shape(self) -> None:
        """reference ``reset_shape``: the solvers (and their captured graphs) are rebuilt on the next solve"""
        self._solvers.clear()

    def reset_cuda_graph(self) -> None:
        self._solvers.clear()

    def update_link_inertial(self, link_name: str, mass: Optional[float] = None, com=None, inertia=None) -> None:
        self.config.kinematics.kinematics_config.update_link_inertial(link_name, mass, com, inertia)

    def update_links_inertial(self, link_properties: Dict) -> None:
        self.config.kinematics.kinematics_config.update_links_inertial(link_properties)

    def update_tool_pose_criteria(self, tool_pose_criteria: Dict) -> None:
        """``{tool frame: ToolPoseCriteria}``: in place on the solvers that exist, and kept for the ones built later"""
        self._criteria = dict(tool_pose_criteria)
        for s in self._solvers.values():
            s.update_tool_pose_criteria(tool_pose_criteria)

    def destroy(self) -> None:
        self._solvers.clear()
        self._checker = None

    def _solver(self, batch: int) -> IKSolver:
        if batch not in self._solvers:
            c = self.config
            cfg = IKSolverCfg(num_seeds=c.num_seeds, position_threshold=c.position_tolerance, rotation_threshold=c.orientation_tolerance,
                              use_lm_seed=c.use_lm_seed, seed_solver_num_seeds=c.seed_solver_num_seeds, num_goalset=c.max_goalset,
                              stream_shards=c.stream_shards if batch % max(c.stream_shards, 1) == 0 else 1,
                              override_iters_for_multi_link_ik=c.override_iters_for_multi_link_ik)
            cfg.rollout.scene_activation_distance = c.optimizer_collision_activation_distance
            if not c.self_collision_check:
                cfg.rollout.self_collision_weight = 0.0
            self._solvers[batch] = IKSolver(c.kinematics.kinematics_config, c.scene, batch, cfg, use_cuda_graph=c.use_cuda_graph)
            if getattr(self, "_criteria", None):
                self._solvers[batch].update_tool_pose_criteria(self._criteria)
        return self._solvers[batch]

    def solve_pose(self, goal_tool_poses: GoalToolPose, seed_config: Optional[torch.Tensor] = None, return_seeds: int = 1,
                   current_state: Optional[JointState] = None, **unused) -> InverseKinematicsResult:
        """``goal_tool_poses``: GoalToolPose [batch, 1, T, num_goalset, 3 | 4] -- the goal of EVERY tool frame of the robot
        (``tool_frames`` order), as the reference's solve_pose takes it (solver_ik.py:631-700); a goal set smaller than
        ``config.max_goalset`` is padded with its last pose, a larger one rebuilds the solvers; ``seed_config``
        [batch, num_seeds, dof] optional warm starts; ``current_state`` [batch, dof]: the robot's configuration -- first seed of
        the seed stage, which then prefers solutions near it (reference ``solve_pose(current_state=)``, solver_ik.py:631-700)."""
        t0 = time.perf_counter()
        gp, gq = goal_tool_poses.static_goals(self.tool_frames)  # [batch, T, G, 3 | 4], the robot's frame order
        B, T, G = int(gp.shape[0]), int(gp.shape[1]), int(gp.shape[2])
        if G > self.config.max_goalset:
            self.config.max_goalset = G
            self._solvers.clear()
        M = self.config.max_goalset
        slv = self._solver(B)
        pos, quat = gp.reshape(B, T, G, 3), gq.reshape(B, T, G, 4)
        if G < M:
            pos = torch.cat([pos, pos[:, :, -1:].expand(B, T, M - G, 3)], 2)
            quat = torch.cat([quat, quat[:, :, -1:].expand(B, T, M - G, 4)], 2)
        cur = None if current_state is None else current_state.position.to(pos.device, torch.float32).reshape(-1, self.dof).expand(B, self.dof)
        r = slv.solve_pose(pos, quat, seeds=seed_config, return_seeds=return_seeds, exit_early=self.config.exit_early, current_position=cur)
        torch.cuda.synchronize(pos.device) if pos.is_cuda else None
        self.solve_time = time.perf_counter() - t0
        k = return_seeds
        sol = r.solution.reshape(B, k, -1)
        return InverseKinematicsResult(
            success=r.success.reshape(B, k), solution=sol, js_solution=JointState.from_position(sol, joint_names=self.joint_names),
            position_error=r.position_error.reshape(B, k), rotation_error=r.rotation_error.reshape(B, k),
            goalset_index=None if r.goalset_index is None else r.goalset_index.reshape(B, k, T), solve_time=self.solve_time,
            debug_info={"optimizer_ran": bool(getattr(slv, "optimizer_ran", True))}, total_time=self.solve_time,
            position_tolerance=self.config.position_tolerance, orientation_tolerance=self.config.orientation_tolerance, batch_size=B,
            num_seeds=self.config.num_seeds)
