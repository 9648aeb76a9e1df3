"""Model-predictive control on the trajectory-optimisation rollout.

Counterpart of the reference's ``MPCSolver`` (``curobo/_src/solver/solver_mpc.py:33-878``; configuration
``solver_mpc_cfg.py:30-287``: ``optimization_dt``, ``interpolation_steps``, ``cold_start_optimization_num_iters``,
``warm_start_optimization_num_iters``): a receding-horizon loop over B-spline knots.

    setup(current_state, goal)              goals (tool pose and / or joint configuration), batch of robots
    optimize_next_action(current_state)     first call: COLD start (more iterations, from a hold-still seed);
                                            then, whenever the command buffer of the current knot interval is used up:
                                            shift the knots by one interval, re-anchor the spline's start state at the
                                            robot's CURRENT position / velocity / acceleration and WARM-start L-BFGS
                                            (fewer iterations) -- solver_mpc.py:533-700
    -> MPCSolverResult.next_action          the next interpolated state (position, velocity, acceleration)

The optimiser iterations are the same hipGraph-captured HIP kernels as trajectory optimisation (fused rollout: pose
tracking along the horizon + c-space STATE limits + self + swept scene collision; line search + two-loop)."""

from __future__ import annotations

import dataclasses
import time
from dataclasses import dataclass, field
from typing import Optional

import torch

from ..optim import LBFGSOpt, LBFGSOptCfg
from ..robot.kinematics_params import KinematicsParams
from ..rollout.trajopt_rollout import TrajOptRollout, TrajOptRolloutCfg
from ..scene.data import SceneData
from ..types import GoalToolPose, JointState


@dataclass
class MPCSolverCfg:
    optimization_dt: float = 0.02          # duration of one knot interval (reference default, solver_mpc_cfg.py:71)
    interpolation_steps: int = 4           # commands per knot interval: command_dt = optimization_dt / interpolation_steps
    n_knots: int = 16
    #: False (the reference): commands are the plan's SECOND knot interval and the plan is renewed every interval from the
    #: robot's measured state -- the command stream then steps by about velocity x optimization_dt at every renewal (the
    #: first interval, which only extrapolates the start state, is skipped).  True: the first TWO intervals of every plan
    #: are executed from sample 1 on, so consecutive plans join with continuous position / velocity / acceleration, at
    #: half the re-planning rate.
    continuous_commands: bool = True
    cold_start_optimization_num_iters: int = 100
    warm_start_optimization_num_iters: int = 25
    #: tracking along the whole horizon; pose weights far below the trajopt task's (1e6 / 1e5: a terminal goal there) so
    #: that the velocity / acceleration bounds of the c-space STATE cost shape the motion at a 5 ms command step
    rollout: TrajOptRolloutCfg = field(default_factory=lambda: TrajOptRolloutCfg(
        non_terminal_pose_factor=1.0, pose_weight=[2000.0, 200.0], cspace_weight=[10000.0, 10000.0, 2000.0, 100.0, 100.0],
        cspace_regularization=[10.0, 100.0, 1.0, 0.0, 10.0],
        # joint-position tracking (off until a goal configuration is given: update_goal_state / update_goal_tool_poses(run_ik=True)),
        # the MPC task's values (lbfgs_mpc.yml:28-29)
        cspace_target_weight=1000.0, cspace_non_terminal_weight_factor=0.05))
    optimizer: LBFGSOptCfg = field(default_factory=lambda: LBFGSOptCfg(history=15, inner_iters=25))
    use_cuda_graph: bool = True
    #: ``prepare_safe_deceleration_trajectory`` (reference solver_mpc_cfg.py:81-90): deceleration seeds for a moving robot, or hold still
    use_deceleration_on_failure: bool = True
    deceleration_profile: str = "exponential"  # "linear" | "exponential" | "smooth"
    #: seeds of the goal IK (update_goal_tool_poses(run_ik=True)): the solution closest to the current configuration is tracked
    goal_ik_seeds: int = 16

    @staticmethod
    def reference_task(**overrides) -> "MPCSolverCfg":
        """The cost weights and optimiser settings of the reference's MPC task file, content/configs/task/mpc/lbfgs_mpc.yml, instead of
        this package's defaults above (which were tuned on this package's own closed-loop tests: lower pose weights, stronger joint-state
        bounds, retimed weights, 15-deep history, 100 / 25 iterations).  The values are held to the file by
        ``tests/test_types_members.py::test_mpc_reference_task_is_the_reference_file``; the closed-loop behaviour of both value sets is
        measured by ``tools/r06/mpc_task_compare.py``."""
        rollout = TrajOptRolloutCfg(
            non_terminal_pose_factor=1.0, pose_weight=[5000.0, 200.0], pose_convergence_tolerance=[0.0, 0.0],
            cspace_weight=[1000.0, 1000.0, 1000.0, 100.0, 0.0], cspace_activation_distance=[0.01] * 5,
            cspace_regularization=[0.01, 10000.0, 10.0, 0.0, 0.0], retime_weights=False, retime_regularization_weights=True,
            cspace_target_weight=1000.0, cspace_non_terminal_weight_factor=0.05,
            scene_activation_distance=0.01, scene_collision_weight=10000.0, use_sweep=True, use_speed_metric=True, self_collision_weight=100000.0)
        optimizer = LBFGSOptCfg(history=27, inner_iters=25, num_iters=50, cost_relative_threshold=1.0, line_search_c_1=1e-3, line_search_c_2=0.98,
                                epsilon=0.01, step_scale=0.98, line_search_scale=[0.0, 0.1, 0.5, 1.0])
        # the iteration counts of the reference's MPCSolverCfg itself (solver_mpc_cfg.py:75-78): 300 for the first solve, 200 for every
        # warm-started one -- eight times this package's 25: with the file's acceleration regularisation (10 000) the short warm
        # starts of the package defaults stall centimetres from the goal (profiles/r06_b_mpc_task_compare.json)
        return MPCSolverCfg(**{**dict(rollout=rollout, optimizer=optimizer, cold_start_optimization_num_iters=300,
                                      warm_start_optimization_num_iters=200), **overrides})


@dataclass
class MPCSolverResult:
    """reference MPCSolverResult (solver_mpc_result.py)"""

    next_action: Optional[JointState] = None
    action_sequence: Optional[JointState] = None   # the remaining planned states [B, steps, D]
    action_buffer: Optional[torch.Tensor] = None   # knots [B, n_knots, D]
    action_dt: float = 0.0
    solve_time: float = 0.0
    position_error: Optional[torch.Tensor] = None  # [B] tool position error of the plan's end point
    rotation_error: Optional[torch.Tensor] = None
    feasible: Optional[torch.Tensor] = None        # [B] no collision / limit violation over the next two knot intervals
    reoptimized: bool = False


class MPCSolver:
    def __init__(self, kin: KinematicsParams, scene: Optional[SceneData], num_robots: int = 1, cfg: Optional[MPCSolverCfg] = None):
        self.kin, self.scene, self.B = kin, scene, num_robots
        from ..scene.data import warn_if_reference_mesh_gradient

        warn_if_reference_mesh_gradient(scene, "MPCSolver")
        self.cfg = cfg or MPCSolverCfg()
        c = self.cfg
        self.device = kin.device
        rc = dataclasses.replace(c.rollout, n_knots=c.n_knots, interpolation_steps=c.interpolation_steps,
                                 traj_dt=c.optimization_dt / c.interpolation_steps)
        self.rollout_cfg = rc
        ocfg = dataclasses.replace(c.optimizer, num_problems=num_robots, num_iters=c.cold_start_optimization_num_iters)
        self.nls = len(ocfg.line_search_scale)
        self.rollout = TrajOptRollout(kin, scene, num_robots * self.nls, rc)
        self.metrics_rollout = TrajOptRollout(kin, scene, num_robots, dataclasses.replace(rc, use_fused=False))
        bounds = (kin.joint_limits_position[0], kin.joint_limits_position[1])
        self.optimizer = LBFGSOpt(ocfg, self.rollout.cost_and_gradient, rc.n_knots, kin.num_dof, bounds, self.device,
                                  use_cuda_graph=c.use_cuda_graph)
        self._rows = torch.arange(num_robots * self.nls, device=self.device, dtype=torch.int32) // self.nls
        self._mrows = torch.arange(num_robots, device=self.device, dtype=torch.int32)
        # goal buffers in their final shape from the start ([B, T, 1, 3 | 4], identity rotations): every later goal update is then
        # written in place, whatever was captured in a hipGraph in between (a re-allocated buffer would be invisible to it)
        T = kin.num_pose_links
        gp0 = torch.zeros(num_robots, T, 1, 3, device=self.device)
        gq0 = torch.zeros(num_robots, T, 1, 4, device=self.device)
        gq0[..., 0] = 1.0
        self.rollout.update_goals(gp0, gq0, self._rows)
        self.metrics_rollout.update_goals(gp0, gq0, self._mrows)
        self._knots: Optional[torch.Tensor] = None
        self._cmd = None       # (position, velocity, acceleration) [B, H, D] of the current plan
        self._cursor = 0
        self._setup_done = False
        self._warm = False

    # ------------------------------------------------------------------ problem definition
    @property
    def command_dt(self) -> float:
        return self.cfg.optimization_dt / self.cfg.interpolation_steps

    def setup(self, current_state: JointState, goal_tool_poses: Optional[GoalToolPose] = None) -> None:
        """first state + goals (reference MPCSolver.setup, solver_mpc.py:261-330)"""
        if goal_tool_poses is not None:
            self.update_goal_tool_poses(goal_tool_poses)
        self.update_current_state(current_state)
        self._setup_done, self._warm = True, False
        self._knots = None

    def update_goal_tool_poses(self, goal_tool_poses: GoalToolPose, run_ik: bool = False, use_ik_goal: bool = True,
                               use_best_effort_ik: bool = False) -> bool:
        """tracked tool poses [B, 1, T, 1, 3 | 4]; may change between control steps (reference :365-438).  ``run_ik``: a
        collision-free IK solution of the goal (seeded with, and preferring solutions close to, the current configuration) becomes
        the goal CONFIGURATION and, with ``use_ik_goal``, joint-position tracking is switched on beside the pose tracking; returns
        False (and leaves the previous goal in place) when the IK fails for a robot, unless ``use_best_effort_ik``.  Without
        ``run_ik`` joint-position tracking is switched off and only the poses are tracked."""
        gp, gq = goal_tool_poses.static_goals(list(self.kin.tool_frames))  # every tool frame, the robot's frame order
        gp, gq = gp.to(self.device, torch.float32).contiguous(), gq.to(self.device, torch.float32).contiguous()
        if run_ik:
            ok, q_goal = self._solve_goal_ik(gp[:, :, :1], gq[:, :, :1])
            if not (use_best_effort_ik or bool(ok.all())):
                return False
            if use_ik_goal:
                self.update_goal_state(JointState.from_position(q_goal, self.kin.joint_names))
                self.enable_joint_position_tracking()
        else:
            self.disable_joint_position_tracking()
        self.rollout.update_goals(gp[:, :, :1], gq[:, :, :1], self._rows)
        self.metrics_rollout.update_goals(gp[:, :, :1], gq[:, :, :1], self._mrows)
        return True

    def _solve_goal_ik(self, gp: torch.Tensor, gq: torch.Tensor):
        """(success [B], configuration [B, D]) of the pose goals (reference _solve_ik_for_goal, :439-456)"""
        from .ik import IKSolver, IKSolverCfg

        if getattr(self, "_ik", None) is None:
            self._ik = IKSolver(self.kin, self.scene, self.B, IKSolverCfg(num_seeds=self.cfg.goal_ik_seeds))
        cur = getattr(self, "_current", None)
        S = self.cfg.goal_ik_seeds
        r = self._ik.solve_pose(gp, gq, current_position=cur, return_seeds=S)
        ok, sol = r.success.reshape(self.B, S), r.solution.reshape(self.B, S, -1)
        if cur is None:
            pick = torch.zeros(self.B, dtype=torch.long, device=self.device)  # (ranked best first)
        else:
            # of the solutions that pass every check, the one CLOSEST to the current configuration: the tracked goal must not
            # send the arm through another branch of the kinematics (the ranking of the IK is by pose error, not by distance)
            far = (sol - cur.view(self.B, 1, -1)).norm(dim=-1)
            pick = torch.where(ok, far, torch.full_like(far, float("inf"))).argmin(dim=1)
        return ok.any(dim=1), sol[torch.arange(self.B, device=self.device), pick]

    def update_goal_state(self, goal_state: JointState) -> None:
        """goal configuration [B, D] of the joint-position tracking term (reference update_goal_state, :458-474); it counts once
        ``enable_joint_position_tracking`` switched the term on"""
        q = goal_state.position.to(self.device, torch.float32).reshape(self.B, self.kin.num_dof)
        self.rollout.update_cspace_target(q, self._rows)
        self.metrics_rollout.update_cspace_target(q, self._mrows)
        self._goal_config = q.clone()

    def enable_joint_position_tracking(self) -> None:
        """reference solver_core.py:404-414"""
        for ro in (self.rollout, self.metrics_rollout):
            ro.enable_cspace_target()

    def disable_joint_position_tracking(self) -> None:
        for ro in (self.rollout, self.metrics_rollout):
            ro.disable_cspace_target()

    def update_current_state(self, current_state: JointState) -> None:
        """the spline starts at the robot's current position / velocity / acceleration (reference :476-497)"""
        D = self.kin.num_dof
        p = current_state.position.to(self.device, torch.float32).reshape(self.B, D)
        v = None if current_state.velocity is None else current_state.velocity.reshape(self.B, D)
        a = None if current_state.acceleration is None else current_state.acceleration.reshape(self.B, D)
        self.rollout.update_start_state(p, v, a, start_idx=self._rows)
        self.metrics_rollout.update_start_state(p, v, a, start_idx=self._mrows)
        self._current = p.clone()

    def reset_robot(self, current_state: JointState) -> None:
        self.update_current_state(current_state)
        self._warm, self._knots, self._seed_override = False, None, None

    # ------------------------------------------------------------------ seeds from outside
    def update_seed_trajectory(self, seed_trajectory: torch.Tensor) -> None:
        """the knots [batch, action_horizon, action_dim] the NEXT solve starts from instead of the hold-still seed (cold) or the
        shifted previous plan (warm); the next ``optimize_next_action`` re-optimises (reference ``update_seed_trajectory``,
        solver_mpc.py:498-514: the action buffer and the optimiser are re-initialised with the trajectory)"""
        nk, D = self.rollout_cfg.n_knots, self.kin.num_dof
        if seed_trajectory.ndim != 3:
            raise ValueError(f"seed_trajectory must have 3 dimensions, got {seed_trajectory.ndim}")
        if seed_trajectory.shape[0] != self.B:
            raise ValueError(f"seed_trajectory must have {self.B} rows, got {seed_trajectory.shape[0]}")
        if seed_trajectory.shape[2] != D:
            raise ValueError(f"seed_trajectory must have {D} columns, got {seed_trajectory.shape[2]}")
        if seed_trajectory.shape[1] != nk:
            raise ValueError(f"seed_trajectory must have {nk} columns, got {seed_trajectory.shape[1]}")
        lo, hi = self.kin.joint_limits_position[0], self.kin.joint_limits_position[1]
        self._seed_override = torch.minimum(torch.maximum(seed_trajectory.to(self.device, torch.float32), lo), hi).contiguous().clone()
        self._cursor = 1 << 30  # (a warm controller has 'used its commands up': the next call solves)

    def update_seed_trajectory_from_goal_state(self, goal_joint_state: JointState) -> None:
        """seed = the straight joint-space line from the current state to ``goal_joint_state`` [batch, dof] over the knots
        (reference :516-531: ``prepare_trajectory_seeds`` with the goal as ``seed_config``)"""
        if getattr(self, "_current", None) is None:
            raise RuntimeError("Current state not available. Call setup first.")
        nk, D = self.rollout_cfg.n_knots, self.kin.num_dof
        goal = goal_joint_state.position.to(self.device, torch.float32).reshape(self.B, 1, D)
        t = torch.linspace(0.0, 1.0, nk + 2, device=self.device)[1:-1].view(1, -1, 1)  # (the weights of TrajOptSolver.seed_knots)
        self.update_seed_trajectory(self._current.view(self.B, 1, D) * (1 - t) + goal * t)

    def prepare_safe_deceleration_trajectory(self, current_state: JointState, failed_mask: torch.Tensor,
                                             deceleration_time: Optional[float] = None, deceleration_profile: Optional[str] = None) -> torch.Tensor:
        """knots [batch, action_horizon, action_dim] that bring the robots to rest from ``current_state`` (position, velocity,
        acceleration [batch, dof]): the reference's deceleration seeds (``util/deceleration.py``) when a robot moves and
        ``cfg.use_deceleration_on_failure``, the hold-still line otherwise.  A utility, as in the reference (solver_mpc.py:701-762; its
        control loop does not call it, :679); hand the result to ``update_seed_trajectory`` to start the next solve from it.
        ``failed_mask`` [batch] and ``deceleration_time`` are accepted for the reference's signature: the trajectory is built for every
        robot (the caller picks the rows), the profile always spans the action horizon."""
        from ..util.deceleration import deceleration_knots

        nk, D = self.rollout_cfg.n_knots, self.kin.num_dof
        p = current_state.position.to(self.device, torch.float32).reshape(self.B, D)
        v = None if current_state.velocity is None else current_state.velocity.to(self.device, torch.float32).reshape(self.B, D)
        if not self.cfg.use_deceleration_on_failure or v is None or not bool((v.abs() > 1e-6).any()):
            return p.view(self.B, 1, D).expand(self.B, nk, D).contiguous()
        a = torch.zeros_like(p) if current_state.acceleration is None else current_state.acceleration.to(self.device, torch.float32).reshape(self.B, D)
        profile = deceleration_profile or self.cfg.deceleration_profile
        return deceleration_knots(p, v, a, float(self.cfg.optimization_dt), nk, profile)

    def _take_seed(self, default: torch.Tensor) -> torch.Tensor:
        seed = getattr(self, "_seed_override", None)
        self._seed_override = None
        return default if seed is None else seed

    # ------------------------------------------------------------------ solves
    def _solve(self, seed_knots: torch.Tensor, iters: int) -> None:
        o = self.optimizer
        o.cfg.num_iters = iters
        best = o.optimize(seed_knots)
        self._knots = best.reshape(self.B, self.rollout_cfg.n_knots, self.kin.num_dof).clone()
        m = self.metrics_rollout
        m.evaluate_action(self._knots, with_gradient=False)
        self._cmd = (m.position.clone(), m.velocity.clone(), m.acceleration.clone())
        # Commands come from the SECOND knot interval (reference: command_start_idx = interpolation_steps for B-spline
        # control spaces, solver_mpc.py:44-55): the first one is spanned by the start state's fixed knots alone (it only
        # extrapolates the current position / velocity / acceleration), the free knots act from the second one on.
        self._cursor = 1 if self.cfg.continuous_commands else self.cfg.interpolation_steps
        H, T = self.rollout_cfg.padded_horizon, self.kin.num_pose_links
        self._pos_err = m.pose_pos_dist.view(self.B, H, T)[:, -1, 0].clone()
        self._rot_err = m.pose_rot_dist.view(self.B, H, T)[:, -1, 0].clone()
        near = 2 * self.cfg.interpolation_steps + 1
        ok = m.self_dist.view(self.B, H)[:, :near].sum(-1) <= 0.0
        if self.scene is not None:
            ok &= m.scene_dist.view(self.B, H, -1)[:, :near].sum((-1, -2)) <= 0.0
        self._feasible = ok

    def cold_start_solve(self, current_state: JointState) -> None:
        """hold-still seed, ``cold_start_optimization_num_iters`` iterations (reference :626-642)"""
        self.update_current_state(current_state)
        seed = self._current.view(self.B, 1, -1).expand(-1, self.rollout_cfg.n_knots, -1).contiguous()
        self._solve(self._take_seed(seed), self.cfg.cold_start_optimization_num_iters)

    def warm_start_solve(self, current_state: JointState) -> None:
        """previous knots shifted by the executed intervals (last knot repeated), ``warm_start_optimization_num_iters`` iterations
        (reference :643-700: trajectory_execution_manager.get_shifted action buffer + optimizer.shift)"""
        self.update_current_state(current_state)
        # shift by the knot intervals the robot actually executed since the last solve: two in continuous mode (the plan is
        # followed from point 1 up to point 2 * interpolation_steps), one otherwise; at least one
        start = 1 if self.cfg.continuous_commands else self.cfg.interpolation_steps
        n = max(1, min((self._cursor - start) // self.cfg.interpolation_steps, self._knots.shape[1] - 1))
        seed = torch.cat([self._knots[:, n:], self._knots[:, -1:].expand(-1, n, -1)], dim=1).contiguous()
        self._solve(self._take_seed(seed), self.cfg.warm_start_optimization_num_iters)

    def optimize_next_action(self, current_state: JointState) -> MPCSolverResult:
        if not self._setup_done:
            raise RuntimeError("MPC problem not setup, call setup first")
        t0 = time.perf_counter()
        reopt = False
        if not self._warm:
            self.cold_start_solve(current_state)
            self._warm, reopt = True, True
        elif self._cursor >= 2 * self.cfg.interpolation_steps + (1 if self.cfg.continuous_commands else 0):  # commands used up
            self.warm_start_solve(current_state)
            reopt = True
        p, v, a = (x[:, self._cursor] for x in self._cmd)
        rest = slice(self._cursor, None)
        self._cursor += 1
        if p.is_cuda:
            torch.cuda.synchronize(p.device)
        return MPCSolverResult(
            next_action=JointState(position=p.clone(), velocity=v.clone(), acceleration=a.clone(), joint_names=self.kin.joint_names),
            action_sequence=JointState(position=self._cmd[0][:, rest], velocity=self._cmd[1][:, rest], acceleration=self._cmd[2][:, rest]),
            action_buffer=self._knots, action_dt=self.command_dt, solve_time=time.perf_counter() - t0, position_error=self._pos_err,
            rotation_error=self._rot_err, feasible=self._feasible, reoptimized=reopt)

    def optimize_action_sequence(self, current_state: JointState) -> MPCSolverResult:
        """always re-optimise and return the whole planned sequence (reference :581-624)"""
        if not self._setup_done:
            raise RuntimeError("MPC problem not setup, call setup first")
        t0 = time.perf_counter()
        (self.warm_start_solve if self._warm else self.cold_start_solve)(current_state)
        self._warm = True
        return MPCSolverResult(action_sequence=JointState(position=self._cmd[0][:, self.cfg.interpolation_steps:], velocity=self._cmd[1][:, self.cfg.interpolation_steps:],
                                                          acceleration=self._cmd[2][:, self.cfg.interpolation_steps:]),
                               action_buffer=self._knots, action_dt=self.command_dt, solve_time=time.perf_counter() - t0,
                               position_error=self._pos_err, rotation_error=self._rot_err, feasible=self._feasible, reoptimized=True)
