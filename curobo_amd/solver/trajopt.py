"""Collision-free trajectory optimisation: B-spline seeds from the start state to per-seed goal
configurations, L-BFGS on the full trajopt rollout of every seed in parallel, time-optimal
finetune passes, best successful seed per problem.

Mirrors the reference ``TrajOptSolver`` (``curobo/_src/solver/solver_trajopt.py``) for the
``lbfgs_bspline_trajopt.yml`` task:

* ``_solve_impl`` (:258-467): seeds (``seed_traj`` / lines to ``seed_config`` / constant,
  ``solver_core.py:136-213``) -> per-seed dt from the seed's own velocity / acceleration / jerk ->
  ``finetune_attempts + 1`` passes, each at ``dt = clamp(best_dt * finetune_dt_scale)`` (:337-348),
  re-seeded from the previous pass's optimised knots (:375-389), followed by the retiming of every
  seed to the fastest dt that respects its limits (``compute_trajectory_dt`` :636-677,
  ``_update_trajectory_dt`` :560-577), the metrics rollout at that dt and the interpolated-trajectory
  check (:399-423); a pass replaces a seed's stored solution when it succeeded at a dt that is not
  slower (:435-447).
* success and ranking (``solver_trajopt_result.py:143-300``): feasible over the horizon (and on the
  interpolated trajectory) and converged at the last point; rank = last-point pose error + 0.001 mean
  |jerk| + 0.01 mean |acc| + 1000 dt, + 1e16 for failures.
* ``solve_pose`` (:679-829) and ``solve_cspace`` (:831-971).

Differences, all on the host side: ``solve_pose`` without ``seed_config`` runs the collision-free IK
itself (the reference leaves that to ``MotionPlanner.plan_pose``, ``curobo_amd.motion_planner`` does the
same composition explicitly); seeds that repeat an earlier seed's goal configuration get a smooth
mid-trajectory bump so that they are distinct.  With ``torch.distributed`` initialised the seed axis is
sharded (``TrajOptSolver.sharded``): every pass runs on the local seeds only, the two host decisions of
the finetune loop are all-reduced so that every world size takes the same passes, and the winner is found
with one all-gather (``curobo_amd.distributed``).
"""

from __future__ import annotations

import dataclasses
import os
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch

from ..distributed import global_topk, shard_range
from ..optim import LBFGSOpt, LBFGSOptCfg
from ..robot.kinematics_params import KinematicsParams
from ..rollout.trajopt_rollout import TrajOptRollout, TrajOptRolloutCfg, joint_limit_vector
from ..scene.data import SceneData
from .ik import IKSolver, IKSolverCfg


#: where the free knots of a straight-line seed sit when a configuration does not say (see TrajOptSolverCfg.seed_knot_placement);
#: CUROBO_SEED_KNOT_PLACEMENT overrides it for A/B measurements
DEFAULT_SEED_KNOT_PLACEMENT = "reference"


@dataclass
class TrajOptSolverCfg:
    num_seeds: int = 4
    position_threshold: float = 0.005
    rotation_threshold: float = 0.05
    seed_bump: float = 0.15  # relative mid-trajectory perturbation of the seeds that repeat a goal configuration
    #: distinct IK solutions the seeds aim at (reference: every trajopt seed gets its own IK solution,
    #: solver_trajopt.py:390-420 / trajectory_seed_generator.py:122-170).  Seed s ends in solution
    #: s % num_ik_goals (the best one when that solution failed); 0 = num_seeds (the reference's
    #: behaviour), 1 = all seeds share the best solution.  Measured (tools/trajopt_goal_diversity.py,
    #: 64 feasible goals, 8 seeds): success 0.95 -> 1.00 (C1 world), 0.89 -> 1.00 (C2 world), same time.
    num_ik_goals: int = 0
    #: goal poses per problem (reference ``max_goalset``): every seed is scored against the closest member of the set, the
    #: result reports which member the winner reached; smaller sets are padded with their last member
    num_goalset: int = 1
    #: ``rollout.traj_dt`` only initialises the dt buffers: every solve sets a dt per seed
    rollout: TrajOptRolloutCfg = field(default_factory=lambda: TrajOptRolloutCfg(traj_dt=0.15))
    optimizer: LBFGSOptCfg = field(default_factory=lambda: LBFGSOptCfg(history=27, inner_iters=25, num_iters=100))
    ik: IKSolverCfg = field(default_factory=lambda: IKSolverCfg(num_seeds=32))
    seed: int = 0
    # reference TrajOptSolverCfg (solver_trajopt_cfg.py:49-77)
    interpolation_dt: float = 0.025
    minimum_trajectory_dt: float = 0.002
    maximum_trajectory_dt: float = 0.2
    #: time-optimal finetune passes after the first solve and the dt factor between passes (solve_pose defaults, :691-697)
    finetune_attempts: int = 1
    finetune_dt_scale: float = 0.55
    #: success also needs the trajectory re-sampled at ``interpolation_dt`` to stay inside the position / velocity /
    #: acceleration / jerk limits and free of self and scene collision (reference interpolated_rollout, :475-497)
    check_interpolated: bool = True
    #: where the free knots of a straight-line seed sit: "reference" (the default since round 6) = linspace(0, 1, n_knots) including
    #: both ends, as util/trajectory_seed_generator.py:16-40 places them (first free knot on the start, last on the goal); "even" =
    #: interior points of linspace(0, 1, n_knots + 2) (this package's placement up to round 5).  Measured on 100 random Franka problems
    #: in two worlds (profiles/r06_b_planner_benchmark_seed_knots_*.json): the same success (100 % / 97 %), the same motion times to
    #: four digits, median plan time 10.0 / 10.9 ms against 10.3 / 10.9 ms -- the optimiser forgets the parametrisation of its seed.
    #: the retime + metrics pass after every optimisation pass replayed from a hipGraph (``TrajOptSolver._metrics_pass``); False = eager
    capture_metrics_pass: bool = field(default_factory=lambda: os.environ.get("CUROBO_CAPTURE_METRICS_PASS", "1") != "0")
    seed_knot_placement: str = field(default_factory=lambda: os.environ.get("CUROBO_SEED_KNOT_PLACEMENT", DEFAULT_SEED_KNOT_PLACEMENT))


@dataclass
class TrajOptResult:
    """``return_seeds`` = 1: one row per problem ([P, ...]); k > 1: [P, k, ...], best first."""

    success: torch.Tensor  # bool
    knots: torch.Tensor  # [.., n_knots, D]
    position: torch.Tensor  # [.., H, D] joint trajectory of the winner at the optimiser's resolution
    position_error: torch.Tensor  # at the last point
    rotation_error: torch.Tensor
    cost: torch.Tensor  # the ranking cost (pose error + smoothness + 1000 dt; + 1e16 when unsuccessful)
    seed_index: torch.Tensor  # global seed index
    goal_config: torch.Tensor  # [.., D] joint configuration the winning seed ends in
    ik_success: Optional[torch.Tensor] = None  # [P] (solve_pose with its own IK only)
    traj_dt: Optional[torch.Tensor] = None  # time step between the points of ``position`` (after retiming)
    velocity: Optional[torch.Tensor] = None
    acceleration: Optional[torch.Tensor] = None
    jerk: Optional[torch.Tensor] = None
    finetune_passes: int = 0  # optimisation passes that ran (1 = no finetune pass)
    implicit_goal: bool = True  # the trajectories end exactly in ``goal_config`` (spline boundary knots)
    goalset_index: Optional[torch.Tensor] = None  # member of the goal set the last point is closest to (first tool frame)
    #: every local seed before ranking: dict(success [P, S], traj_dt [P, S], knots [P, S, n_knots, D], cost [P, S])
    all_seeds: Optional[dict] = None

    @property
    def motion_time(self) -> torch.Tensor:
        """duration of the winner: (points - 1) x dt"""
        return (self.position.shape[-2] - 1) * self.traj_dt


class TrajOptSolver:
    def __init__(self, kin: KinematicsParams, scene: Optional[SceneData], num_problems: int,
                 cfg: Optional[TrajOptSolverCfg] = None, use_cuda_graph: bool = True, seed_offset: int = 0,
                 global_num_seeds: Optional[int] = None):
        """``cfg.num_seeds`` seeds per problem run in this process: seeds ``[seed_offset, seed_offset + num_seeds)`` of
        ``global_num_seeds`` when the seed axis is sharded over ranks (``TrajOptSolver.sharded``)."""
        self.kin, self.scene, self.cfg = kin, scene, cfg or TrajOptSolverCfg()
        from ..scene.data import warn_if_reference_mesh_gradient

        warn_if_reference_mesh_gradient(scene, "TrajOptSolver")
        self.P, self.S, self.device = num_problems, self.cfg.num_seeds, kin.device
        self.seed_offset = int(seed_offset)
        self.S_global = int(global_num_seeds) if global_num_seeds is not None else self.S
        # private copy of the optimiser configuration: the caller's cfg may be shared by solvers of other sizes
        ocfg = dataclasses.replace(self.cfg.optimizer, num_problems=self.P * self.S)
        self.cfg = dataclasses.replace(self.cfg, optimizer=ocfg)
        self.nls = len(ocfg.line_search_scale)
        rc = self.cfg.rollout
        self._use_graph = use_cuda_graph
        self._ik: Optional[IKSolver] = None  # built on first use: callers that bring seed_config never need it
        self.rollout = TrajOptRollout(kin, scene, self.P * self.S * self.nls, rc)
        # the metrics rollout checks feasibility the way the reference's does (content/configs/task/metrics_base.yml:8-19):
        # discrete scene collision and the joint-state limits at zero activation distance (a seed fails when a sphere penetrates or a
        # limit is crossed, not when it enters the optimiser's 2.5 mm / 0.01 activation shells), kernel sequence (it materialises the
        # per-point terms the checks read)
        self.metrics_rollout = TrajOptRollout(kin, scene, self.P * self.S, dataclasses.replace(
            rc, use_sweep=False, use_speed_metric=False, scene_activation_distance=0.0, use_fused=False,
            cspace_activation_distance=[0.0] * len(rc.cspace_activation_distance)))
        self.K = max(1, min(self.cfg.num_ik_goals or self.S_global, self.S_global, self.cfg.ik.num_seeds))
        D, PS = kin.num_dof, self.P * self.S
        rows = torch.arange(PS * self.nls, device=self.device)
        self._row_seed = (rows // self.nls).to(torch.int32)  # (problem, seed) row of every optimiser trajectory
        self._row_problem = (rows // (self.S * self.nls)).to(torch.int32)
        self._mrow_seed = torch.arange(PS, device=self.device, dtype=torch.int32)
        self._mrow_problem = (torch.arange(PS, device=self.device) // self.S).to(torch.int32)
        # one goal joint state (and one dt) per (problem, seed); one start state per problem: allocated before any capture
        for r, seed_rows, prob_rows in ((self.rollout, self._row_seed, self._row_problem),
                                        (self.metrics_rollout, self._mrow_seed, self._mrow_problem)):
            r.update_start_state(torch.zeros(self.P, D, device=self.device), start_idx=prob_rows)
            r.update_goal_state(torch.zeros(PS, D, device=self.device), seed_rows)
        bounds = (kin.joint_limits_position[0], kin.joint_limits_position[1])
        self.optimizer = LBFGSOpt(ocfg, self.rollout.cost_and_gradient, rc.n_knots, kin.num_dof, bounds, self.device,
                                  use_cuda_graph=use_cuda_graph)
        self.optimizer.rank_sharded = self.S_global != self.S  # seeds split over ranks: the convergence exit is a collective
        self._check = _InterpolatedCheck(kin, scene, rc) if self.cfg.check_interpolated else None
        self._pass_graphs: dict = {}  # captured metrics passes, per goal kind (_metrics_pass)

    @classmethod
    def sharded(cls, kin: KinematicsParams, scene: Optional[SceneData], num_problems: int,
                cfg: Optional[TrajOptSolverCfg] = None, use_cuda_graph: bool = True) -> "TrajOptSolver":
        """``cfg.num_seeds`` is the GLOBAL seed count; the solver of this rank optimises its contiguous shard of it.
        Alone in the process this is the plain solver."""
        import torch.distributed as dist

        cfg = cfg or TrajOptSolverCfg()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return cls(kin, scene, num_problems, cfg, use_cuda_graph=use_cuda_graph)
        lo, hi = shard_range(cfg.num_seeds, dist.get_rank(), dist.get_world_size())
        if hi == lo:
            raise ValueError(f"{cfg.num_seeds} seeds cannot be sharded over {dist.get_world_size()} ranks")
        return cls(kin, scene, num_problems, dataclasses.replace(cfg, num_seeds=hi - lo), use_cuda_graph=use_cuda_graph,
                   seed_offset=lo, global_num_seeds=cfg.num_seeds)

    @property
    def ik(self) -> IKSolver:
        if self._ik is None:
            self._ik = IKSolver.sharded(self.kin, self.scene, self.P, dataclasses.replace(self.cfg.ik, num_goalset=self.cfg.num_goalset),
                                        use_cuda_graph=self._use_graph)
            if getattr(self, "_criteria", None):
                self._ik.update_tool_pose_criteria(self._criteria)
        return self._ik

    def reset_seed(self) -> None:
        """reference ``reset_seed``: the next solve draws the seeds the first one drew (the LM seed stage of the IK samples
        its Halton points from a stream that otherwise runs on from solve to solve)"""
        if self._ik is not None:
            self._ik.reset_seed()

    def update_tool_pose_criteria(self, criteria) -> None:
        """``{tool frame: ToolPoseCriteria}`` (reference TrajOptSolver.update_tool_pose_criteria): the optimiser's and the
        metrics rollout's pose-cost rows are rewritten in place (captured graphs read the new values)"""
        self._criteria = dict(criteria)
        for r in (self.rollout, self.metrics_rollout):
            r.update_tool_pose_criteria(criteria)
        if self._ik is not None:
            self._ik.update_tool_pose_criteria(criteria)

    # ------------------------------------------------------------------ seeds
    def seed_goal_choice(self, ik_success: torch.Tensor) -> torch.Tensor:
        """[P, S_global] index (0..K-1) of the IK solution seed s of problem p ends in: s % K when that
        solution passed the IK checks, else the best one (solutions are ranked best first)."""
        SG = getattr(self, "S_global", self.S)
        s_goal = (torch.arange(SG, device=self.device) % self.K).view(1, SG).expand(self.P, SG)
        ok = torch.gather(ik_success.view(self.P, self.K), 1, s_goal)
        return torch.where(ok, s_goal, torch.zeros_like(s_goal))

    def seed_knots(self, start: torch.Tensor, goal_config: torch.Tensor, choice: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[P, S_global, n_knots, D]: straight joint-space lines start -> the seed's goal configuration
        (reference seed generation, solver_trajopt.py:390-420).  Knot placement as the reference's generator has it by default
        (util/trajectory_seed_generator.py:16-40: weights linspace(0, 1, n_knots) INCLUDING both ends, so the first free knot
        repeats the start and the last the goal -- a line that leaves and arrives slowly); ``seed_knot_placement = "even"`` puts
        the free knots on the interior points of linspace(0, 1, n_knots + 2) instead.  Both are the same straight line in joint
        space; only seeds differ, not costs.  ``start`` [1 or P, D], ``goal_config`` [P, K, D] (or [P, D]), ``choice``
        [P, S_global] from ``seed_goal_choice``; a seed that repeats an earlier seed's goal adds a smooth
        mid-trajectory bump so that the seeds stay distinct.  Every rank builds the global set (host generator)."""
        rc, D, P, S = self.cfg.rollout, self.kin.num_dof, self.P, getattr(self, "S_global", self.S)
        goal_config = goal_config.reshape(P, -1, D)
        if choice is None:
            choice = torch.zeros(P, S, dtype=torch.int64, device=self.device)
        goal = torch.gather(goal_config, 1, choice.unsqueeze(-1).expand(P, S, D))  # [P, S, D]
        if self.cfg.seed_knot_placement == "reference":
            t = torch.linspace(0.0, 1.0, rc.n_knots, device=self.device).view(1, 1, -1, 1)
        elif self.cfg.seed_knot_placement == "even":
            t = torch.linspace(0.0, 1.0, rc.n_knots + 2, device=self.device)[1:-1].view(1, 1, -1, 1)
        else:
            raise ValueError(f"seed_knot_placement must be 'even' or 'reference', got {self.cfg.seed_knot_placement!r}")
        line = start.reshape(-1, 1, 1, D) * (1 - t) + goal.view(P, S, 1, D) * t
        gen = torch.Generator(device="cpu").manual_seed(self.cfg.seed)
        half = 0.5 * (self.kin.joint_limits_position[1] - self.kin.joint_limits_position[0])
        bump = torch.randn(P, S, 1, D, generator=gen).to(self.device) * self.cfg.seed_bump * half
        sidx = torch.arange(S, device=self.device).view(1, S)
        first_use = (choice == sidx) & (sidx < goal_config.shape[1])  # seed s is the first one aimed at goal s
        bump = bump * (~first_use).view(P, S, 1, 1)
        knots = line + bump * torch.sin(torch.pi * t)
        lo, hi = self.kin.joint_limits_position[0], self.kin.joint_limits_position[1]
        return torch.minimum(torch.maximum(knots, lo + 1e-3), hi - 1e-3).contiguous()

    def _local(self, x: torch.Tensor) -> torch.Tensor:
        """[P, S_global, ...] -> this rank's [P, S, ...]"""
        return x if x.shape[1] == self.S else x[:, self.seed_offset:self.seed_offset + self.S].contiguous()

    def prepare_trajectory_seeds(self, start: torch.Tensor, seed_config: Optional[torch.Tensor],
                                 seed_traj: Optional[torch.Tensor]) -> torch.Tensor:
        """reference ``prepare_trajectory_seeds`` (solver_core.py:136-213): given trajectories first, then straight lines
        start -> ``seed_config`` (first knot = start, last knot = the configuration, ``TrajectorySeedGenerator``), else
        constant seeds at the start; [P, S_global (or S), n_knots, D]"""
        from ..util.knot_seeds import TrajectorySeedGenerator

        rc, D, P, S = self.cfg.rollout, self.kin.num_dof, self.P, self.S_global
        gen = TrajectorySeedGenerator(rc.n_knots, D, device=self.device)
        start = start.reshape(-1, D).expand(P, D)
        parts, left = [], S
        if seed_traj is not None:
            seed_traj = seed_traj.to(self.device, torch.float32)
            if seed_traj.ndim != 4 or seed_traj.shape[0] != P or tuple(seed_traj.shape[2:]) != (rc.n_knots, D):
                raise ValueError(f"Invalid seed_traj shape {tuple(seed_traj.shape)}. Expected ({P}, n, {rc.n_knots}, {D})")
            use = min(seed_traj.shape[1], left)
            parts.append(seed_traj[:, :use])
            left -= use
        if left > 0:
            if seed_config is not None:
                seed_config = seed_config.to(self.device, torch.float32).reshape(P, -1, D)
                if seed_config.shape[1] < left:
                    raise ValueError(f"Insufficient seed configs: {seed_config.shape[1]} provided, {left} needed")
                parts.append(gen.generate_interpolated_seeds(start, seed_config[:, :left].contiguous(), left))
            else:
                parts.append(gen.generate_constant_seeds(start, left))
        return torch.cat(parts, dim=1).contiguous()

    # ------------------------------------------------------------------ dt
    def compute_trajectory_dt(self, velocity: torch.Tensor, acceleration: torch.Tensor, jerk: torch.Tensor,
                              dt: Optional[torch.Tensor] = None, epsilon: float = 1e-3) -> torch.Tensor:
        """reference TrajOptSolver.compute_trajectory_dt (:636-677): the dt at which the trajectory just
        respects the velocity / acceleration / jerk limits, clamped to [minimum, maximum]_trajectory_dt;
        inputs [B, H, D] sampled at ``dt`` [B] (or scalar ``cfg.rollout.traj_dt``)."""
        from ..util.trajectory import calculate_dt_no_clamp

        rc, D = self.cfg.rollout, self.kin.num_dof
        vmax = self.kin.joint_limits_velocity[1].abs()
        amax, jmax = joint_limit_vector(rc.max_acceleration, D, self.device), joint_limit_vector(rc.max_jerk, D, self.device)  # per joint
        score = calculate_dt_no_clamp(velocity, acceleration, jerk, vmax, amax, jmax, epsilon)
        base = dt if dt is not None else torch.full_like(score, rc.traj_dt)
        return torch.clamp(score * base, min=self.cfg.minimum_trajectory_dt, max=self.cfg.maximum_trajectory_dt)

    def _set_dt(self, dt: torch.Tensor) -> None:
        """dt [P * S] of every (problem, seed) into both rollouts (reference _update_trajectory_dt, :560-577).  On a seed shard
        this is a COLLECTIVE (rank 0's dt of global trajectory 0 is broadcast): every rank must reach it the same number of
        times, i.e. solve arguments that steer the host-side control flow (dt given or not, finetune_attempts, iteration counts)
        must be identical on all ranks -- as must the decisions taken through ``_any``, which are all-reduced for that reason."""
        speed = None
        if self.S_global != self.S:  # the speed metric reads the dt of global trajectory 0 (wp_speed_metric.py:54): rank 0's
            from ..distributed import broadcast_from_rank0

            speed = broadcast_from_rank0(dt[:1].clone())
        self.rollout.update_traj_dt(dt, speed)
        self.metrics_rollout.update_traj_dt(dt, speed)

    # ------------------------------------------------------------------ solve
    def solve_pose(self, start_position: torch.Tensor, goal_position: torch.Tensor, goal_quat: torch.Tensor,
                   env_idx: Optional[torch.Tensor] = None, seed_config: Optional[torch.Tensor] = None,
                   seed_traj: Optional[torch.Tensor] = None, return_seeds: int = 1, dt: Optional[torch.Tensor] = None,
                   use_implicit_goal: bool = True, finetune_attempts: Optional[int] = None,
                   goal_state: Optional[torch.Tensor] = None, initial_iters: Optional[int] = None,
                   time_optimal_iters: Optional[int] = None, finetune_iters: Optional[int] = None,
                   finetune_dt_scale: Optional[float] = None) -> TrajOptResult:
        """``start_position`` [D] (shared) or [P, D]; goal_position [P, 3], goal_quat [P, 4] (wxyz) -- or [P, T, 3 | 4] per
        tool frame, or [P, T, g, 3 | 4] with a goal set of g <= ``cfg.num_goalset`` poses per frame; ``env_idx``
        [P]: problem p plans in scene environment env_idx[p] (reference batch-env planning,
        motion_planner_batch.py; ``idxs_env`` / ``use_multi_env`` of the collision costs).  ``seed_config``
        [P, n >= num_seeds, D]: goal configurations of the seeds (reference: the IK solutions the planner passes,
        motion_planner.py:262-284); without it (and without ``seed_traj``) the solver runs the collision-free IK
        itself.  ``use_implicit_goal``: seed s ends at rest exactly in its goal configuration (the spline's goal
        boundary knots; ``goal_state`` [P, D] overrides the configuration).  Remaining arguments as the reference's
        (:679-829); ``finetune_*`` default to the configuration."""
        P, D = self.P, self.kin.num_dof
        start = start_position.to(self.device, torch.float32).reshape(-1, D)
        ik_ok = None
        if seed_config is None and seed_traj is None:
            # the IK stage with the IK solver's own configuration (exit_early = True by default): when the Levenberg-Marquardt
            # seed stage already solves every problem its L-BFGS stage is skipped, as in the reference's planner
            # (motion_planner.py:249-253 calls ik_solver.solve_pose with the configured exit_early; it is switched off only
            # for the warm-up, :143-144 and :179, so that the optimiser's graph gets captured)
            K = self.K
            gp_ik, gq_ik = self._goal_sets(goal_position, goal_quat)
            # (every tool frame's goal set; the start configuration goes along: goal configurations near it rank first)
            ikr = self.ik.solve_pose(gp_ik, gq_ik, return_seeds=K, env_idx=env_idx, current_position=start.expand(P, D).contiguous())
            ik_ok = ikr.success.view(P, K)
            ik_q = ikr.solution.reshape(P, K, D).contiguous()
            choice = self.seed_goal_choice(ik_ok)  # [P, S_global]
            seeds = self.seed_knots(start, ik_q, choice)
            seed_goal = torch.gather(ik_q, 1, choice.unsqueeze(-1).expand(P, self.S_global, D))
            ik_ok = ik_ok[:, 0]
        else:
            seeds = self.prepare_trajectory_seeds(start, seed_config, seed_traj)
            seed_goal = seeds[:, :, -1, :]  # reference :293-296: the seed's last knot
        if goal_state is not None:
            seed_goal = goal_state.to(self.device, torch.float32).reshape(P, 1, D).expand(P, self.S_global, D)
        res = self._solve_impl(start, goal_position, goal_quat, env_idx, self._local(seeds), self._local(seed_goal.contiguous()),
                               use_implicit_goal, return_seeds, dt, finetune_attempts, initial_iters, time_optimal_iters,
                               finetune_iters, finetune_dt_scale)
        res.ik_success = ik_ok
        return res

    def solve_cspace(self, start_position: torch.Tensor, goal_state: torch.Tensor, env_idx: Optional[torch.Tensor] = None,
                     seed_traj: Optional[torch.Tensor] = None, return_seeds: int = 1, dt: Optional[torch.Tensor] = None,
                     finetune_attempts: Optional[int] = None, initial_iters: Optional[int] = None,
                     time_optimal_iters: Optional[int] = None, finetune_iters: Optional[int] = None,
                     finetune_dt_scale: Optional[float] = None) -> TrajOptResult:
        """Joint-space goal (reference ``solve_cspace``, :831-971): the tool-pose target is the forward kinematics of
        ``goal_state`` [P, D], every seed is the line to it and ends exactly there (implicit goal state)."""
        P, D = self.P, self.kin.num_dof
        goal = goal_state.to(self.device, torch.float32).reshape(P, D).contiguous()
        gp, gq = self._tool_pose_of(goal)
        start = start_position.to(self.device, torch.float32).reshape(-1, D)
        if seed_traj is None:
            # the reference repeats one line for every seed; here the repeats are bumped so that the seeds differ
            seeds = self.seed_knots(start, goal.view(P, 1, D), None)
        else:
            seeds = self.prepare_trajectory_seeds(start, goal.view(P, 1, D).expand(P, self.S_global, D), seed_traj)
        seed_goal = goal.view(P, 1, D).expand(P, self.S_global, D).contiguous()
        return self._solve_impl(start, gp, gq, env_idx, self._local(seeds), self._local(seed_goal), True, return_seeds, dt,
                                finetune_attempts, initial_iters, time_optimal_iters, finetune_iters, finetune_dt_scale)

    def _tool_pose_of(self, q: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """forward kinematics of q [n, D] -> tool-frame position [n, T, 3] and quaternion [n, T, 4]"""
        from ..backends import kinematics as kinematics_hip

        k, n, dev = self.kin, q.shape[0], self.device
        T, S, L = k.num_pose_links, k.num_spheres, k.num_links
        pos, quat = torch.zeros(n, 1, T, 3, device=dev), torch.zeros(n, 1, T, 4, device=dev)
        sph, com, cumul = torch.zeros(n, 1, max(S, 1), 4, device=dev), torch.zeros(n, 1, 4, device=dev), torch.zeros(n, 1, L, 3, 4, device=dev)
        env = torch.zeros(n, dtype=torch.int32, device=dev)
        kinematics_hip.launch_kinematics_forward_spheres(
            pos, quat, sph, com, cumul, q.contiguous(), k.fixed_transforms, k.link_spheres, k.link_masses_com, k.joint_map_type,
            k.joint_map, k.link_map, k.tool_frame_map, k.link_sphere_idx_map, k.joint_offset_map, env, k.num_envs, n, 1,
            k.num_dof, S, 32, True, False)
        return pos.view(n, T, 3), quat.view(n, T, 4)

    def _goal_sets(self, goal_position: torch.Tensor, goal_quat: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """[P, 3] | [P, T, 3] | [P, T, g, 3] (and the quaternions) -> [P, T, cfg.num_goalset, 3 | 4]; a set smaller than the
        solver's is padded with its last member (the closest-member search then never prefers the padding: ties go to the
        lower index)"""
        P, T, G = self.P, self.kin.num_pose_links, self.cfg.num_goalset
        out = []
        for x, w in ((goal_position, 3), (goal_quat, 4)):
            x = x.to(self.device, torch.float32)
            if x.ndim <= 2 or x.numel() == P * w:
                x = x.reshape(P, 1, 1, w).expand(P, T, 1, w)
            elif x.ndim == 3:
                x = x.reshape(P, T, 1, w)
            g = x.shape[2]
            if g > G:
                raise ValueError(f"goal set of {g} poses exceeds the solver's num_goalset ({G})")
            if g < G:
                x = torch.cat([x, x[:, :, -1:].expand(P, T, G - g, w)], 2)
            out.append(x.contiguous())
        return out[0], out[1]

    def _set_problem(self, start, goal_position, goal_quat, env_idx, seed_goal, use_implicit_goal) -> None:
        P, S, D, T = self.P, self.S, self.kin.num_dof, self.kin.num_pose_links
        mode = env_idx is not None
        if mode != getattr(self, "_env_mode", False):
            self.optimizer._graph = None  # the multi-env flag is a kernel argument: capture again
        self._env_mode = mode
        env = env_idx.to(self.device).long().view(P) if mode else None
        gp, gq = self._goal_sets(goal_position, goal_quat)
        startP = start.expand(P, D).contiguous()
        for r, seed_rows, prob_rows in ((self.rollout, self._row_seed, self._row_problem),
                                        (self.metrics_rollout, self._mrow_seed, self._mrow_problem)):
            r.update_start_state(startP, start_idx=prob_rows)
            r.update_env_query_idx(env[prob_rows.long()] if mode else None)
            r.update_goals(gp, gq, prob_rows)
            # end at rest in the seed's goal configuration (implicit goal state); otherwise a free end point at rest
            r.update_goal_state(seed_goal.reshape(P * S, D), seed_rows, implicit=use_implicit_goal)

    def _solve_impl(self, start, goal_position, goal_quat, env_idx, seeds, seed_goal, use_implicit_goal, return_seeds, dt,
                    finetune_attempts, initial_iters, time_optimal_iters, finetune_iters, finetune_dt_scale) -> TrajOptResult:
        cfg, rc = self.cfg, self.cfg.rollout
        P, S, D, nk = self.P, self.S, self.kin.num_dof, rc.n_knots
        finetune_attempts = cfg.finetune_attempts if finetune_attempts is None else int(finetune_attempts)
        scale = cfg.finetune_dt_scale if finetune_dt_scale is None else float(finetune_dt_scale)
        if return_seeds > self.S_global:
            raise ValueError(f"return_seeds ({return_seeds}) exceeds num_seeds ({self.S_global})")
        self._set_problem(start, goal_position, goal_quat, env_idx, seed_goal, use_implicit_goal)
        m = self.metrics_rollout
        action_seed = seeds.reshape(P * S, nk, D).contiguous()
        # dt of every seed: given, or the fastest one its own velocity / acceleration / jerk allow (reference :304-335:
        # the seed is sampled at dt = 1 and scaled)
        if dt is None:
            one = torch.ones(P * S, device=self.device)
            self._set_dt(one)
            m.compute_state_from_action(action_seed)
            best_dt = self.compute_trajectory_dt(m.velocity, m.acceleration, m.jerk, one)
        else:
            best_dt = self._local(torch.as_tensor(dt, dtype=torch.float32, device=self.device).reshape(P, -1).expand(P, self.S_global)).reshape(P * S).clone()
        best = None
        passes = 0
        self.last_pass_trace = [dict(seed_dt=best_dt.view(P, S).clone())]  # per pass: dt it ran at, dt after retiming, successes
        for i in range(finetune_attempts + 1):
            cur_dt = torch.clamp(best_dt * scale, min=cfg.minimum_trajectory_dt, max=cfg.maximum_trajectory_dt)
            self._set_dt(cur_dt)
            iters = initial_iters if i == 0 else (time_optimal_iters if i == 1 else finetune_iters)
            cur_seed = action_seed if i == 0 else best["knots"].reshape(P * S, nk, D).clone()
            opt = self.optimizer.optimize(cur_seed, num_iters=iters)
            passes += 1
            knots = opt.reshape(P * S, nk, D).contiguous()
            # retime every seed to the fastest dt its optimised trajectory allows, then the metrics at that dt
            new_dt, r = self._metrics_pass(knots, cur_dt, start, seed_goal, use_implicit_goal)
            self.last_pass_trace.append(dict(run_dt=cur_dt.view(P, S).clone(), retimed_dt=new_dt.view(P, S).clone(),
                                             success=r["success"].view(P, S).clone()))
            if best is None:
                best, best_dt = r, new_dt.clone()
            else:
                update = r["success"] & (new_dt <= best_dt)
                if not self._any(update):
                    break
                for key, val in r.items():
                    best[key] = torch.where(update.view(-1, *([1] * (val.ndim - 1))), val, best[key])
                best_dt = best["dt"].clone()
            if i == 0 and not self._any(best["success"]):
                break
        res = self._rank(best, seed_goal, return_seeds, passes)
        res.implicit_goal = bool(use_implicit_goal)
        return res

    def _metrics_pass_eager(self, knots, cur_dt, start, seed_goal, use_implicit_goal, static_steps=None):
        m = self.metrics_rollout
        m.compute_state_from_action(knots)
        new_dt = self.compute_trajectory_dt(m.velocity, m.acceleration, m.jerk, cur_dt)
        self._set_dt(new_dt)
        return new_dt, self._seed_metrics(knots, new_dt, start, seed_goal, use_implicit_goal, static_steps)

    def _metrics_pass(self, knots, cur_dt, start, seed_goal, use_implicit_goal):
        """State from the knots -> retimed dt -> metrics rollout -> per-seed success and rank cost (reference solver_trajopt.py:
        469-484 + rollout/metrics.py:233-265).  Run eagerly the pass is ~360 small launches (torch element-wise kernels around the
        metrics rollout's), ~2 ms of host time per pass at ~6 us each; it has no data-dependent shape once the interpolated check
        samples for the LONGEST trajectory the dt range allows instead of reading the longest one back, so it is captured once
        per (solver, goal kind) into a hipGraph and replayed.  Inputs are copied into the graph's buffers, outputs are cloned out
        of them (the next pass overwrites them).  Not on a seed shard (``_set_dt`` is a collective there) and not without graphs."""
        if not (self._use_graph and self.S_global == self.S and knots.is_cuda and self.cfg.capture_metrics_pass):
            return self._metrics_pass_eager(knots, cur_dt, start, seed_goal, use_implicit_goal)
        key = bool(use_implicit_goal)
        g = self._pass_graphs.get(key)
        if g is None:
            rc, cfg = self.cfg.rollout, self.cfg
            # samples of the longest trajectory: knot spacing maximum_trajectory_dt x interpolation_steps (calculate_traj_steps)
            per = int((cfg.maximum_trajectory_dt * rc.interpolation_steps + cfg.interpolation_dt) / cfg.interpolation_dt)
            static_steps = -(-((rc.n_knots + rc.bspline_degree + 1) * per + 1) // 32) * 32
            # (one start row per problem in the graph's buffer, whatever shape the first caller had: [1, D] and [P, D] both fit)
            buf = dict(knots=knots.clone(), cur_dt=cur_dt.clone(), start=start.to(self.device).reshape(-1, self.kin.num_dof).expand(self.P, -1).clone(),
                       seed_goal=seed_goal.clone())
            run = lambda: self._metrics_pass_eager(buf["knots"], buf["cur_dt"], buf["start"], buf["seed_goal"], use_implicit_goal, static_steps)  # noqa: E731
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up off the capture (allocations, lazily built tables)
                run()
                run()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out_dt, out = run()
            g = self._pass_graphs[key] = dict(graph=graph, buf=buf, out_dt=out_dt, out=out)
        b = g["buf"]
        b["knots"].copy_(knots); b["cur_dt"].copy_(cur_dt); b["seed_goal"].copy_(seed_goal.reshape(b["seed_goal"].shape))
        b["start"].copy_(start.to(self.device).reshape(-1, self.kin.num_dof).expand_as(b["start"]))
        g["graph"].replay()
        return g["out_dt"].clone(), {k: v.clone() for k, v in g["out"].items()}

    @staticmethod
    def _any(mask: torch.Tensor) -> bool:
        """``mask.any()`` over the seeds of ALL ranks (the host decisions of the finetune loop, reference :441-450, are taken
        on the global seed set so that every world size runs the same passes)"""
        from ..distributed import all_reduce_max

        return bool(all_reduce_max(mask.any().to(torch.int32).view(1)).item())

    def _seed_metrics(self, knots, dt, start, seed_goal, use_implicit_goal, static_steps=None) -> dict:
        """metrics rollout of P * S optimised seeds at their dt: success = feasible over the horizon (and on the
        interpolated trajectory) and converged at the last point (reference _process_metrics, solver_trajopt_result.py:
        143-238); rank cost of _jit_compute_rank (:271-300)"""
        cfg, rc = self.cfg, self.cfg.rollout
        P, S, D, T = self.P, self.S, self.kin.num_dof, self.kin.num_pose_links
        m = self.metrics_rollout
        m.evaluate_action(knots, with_gradient=False)
        pos_err = m.pose_pos_dist.view(P * S, -1, T)[:, -1].amax(-1)
        rot_err = m.pose_rot_dist.view(P * S, -1, T)[:, -1].amax(-1)
        goalset_index = m.goalset_idx.view(P * S, -1, T)[:, -1, 0].clone()
        lo, hi = self.kin.joint_limits_position[0], self.kin.joint_limits_position[1]
        q = m.position
        feasible = ((q >= lo - 1e-4) & (q <= hi + 1e-4)).all(-1).all(-1)
        # velocity / acceleration / jerk inside their limits at the seed's dt: the reference's success mask is the
        # c-space STATE constraint at zero activation distance over position, velocity, acceleration, jerk (and torque)
        # (content/configs/task/metrics_base.yml:16-19, solver/solver_trajopt_result.py:154-210)
        for x, b in ((m.velocity, m._v_b), (m.acceleration, m._a_b), (m.jerk, m._j_b)):
            feasible &= ((x >= b[0] - 1e-3 * b[0].abs() - 1e-4) & (x <= b[1] + 1e-3 * b[1].abs() + 1e-4)).all(-1).all(-1)
        in_limits = feasible.clone()
        no_self = m.self_dist.view(P * S, -1).sum(-1) <= 0.0
        no_scene = (m.scene_dist.view(P * S, -1).sum(-1) <= 0.0) if self.scene is not None else torch.ones_like(no_self)
        feasible = feasible & no_self & no_scene
        if rc.use_torque_limits:  # inverse-dynamics torques of the whole trajectory inside the effort limits
            feasible &= (m._tau.view(P * S, -1, D).abs() <= m._effort_b[1] * (1.0 + 1e-3) + 1e-3).all(-1).all(-1)
        ok_rollout = feasible.clone()
        if self._check is not None:
            env = m.env_query_idx if m.use_multi_env else None
            feasible &= self._check.feasible(knots, dt, start.expand(P, D).contiguous(), self._mrow_problem, seed_goal.reshape(P * S, D),
                                             use_implicit_goal, cfg.interpolation_dt, env, static_steps=static_steps)
        converged = (pos_err < cfg.position_threshold) & (rot_err < cfg.rotation_threshold)
        ok = feasible & converged
        H = q.shape[1]
        st, en = (8, H - 8) if H > 17 else (0, H)
        mean_jerk = m.jerk.abs().mean(-1)[:, st:en].mean(-1)
        mean_acc = m.acceleration.abs().mean(-1)[:, st:en].mean(-1)
        rank = pos_err + rot_err + 0.001 * mean_jerk + 0.01 * mean_acc + 1000.0 * dt
        return dict(knots=knots.clone(), dt=dt.clone(), success=ok, pos_err=pos_err.clone(), rot_err=rot_err.clone(),
                    rank=rank, position=q.clone(), velocity=m.velocity.clone(), acceleration=m.acceleration.clone(),
                    jerk=m.jerk.clone(), feasible_rollout=ok_rollout, feasible_interpolated=feasible.clone(), converged=converged,
                    in_limits=in_limits, no_self_collision=no_self, no_scene_collision=no_scene, goalset_index=goalset_index)

    def _rank(self, best: dict, seed_goal: torch.Tensor, k: int, passes: int) -> TrajOptResult:
        """the k best seeds per problem over ALL ranks: one all-gather of (rank cost, global seed index, payload)"""
        rc = self.cfg.rollout
        P, S, D, nk = self.P, self.S, self.kin.num_dof, rc.n_knots
        H = best["position"].shape[1]
        ranked = (best["rank"] + 1e16 * (~best["success"]).float()).view(P, S)
        f = lambda x: x.reshape(P, S, -1)  # noqa: E731
        parts = [f(best["knots"]), f(best["position"]), f(best["velocity"]), f(best["acceleration"]), f(best["jerk"]),
                 f(seed_goal), f(best["pos_err"]), f(best["rot_err"]), f(best["success"].float()), f(best["dt"]),
                 f(best["goalset_index"].float())]
        payload = torch.cat(parts, dim=-1)
        cost, idx, win = global_topk(ranked, payload, self.seed_offset, k)
        sizes = [p.shape[-1] for p in parts]
        kn, pos, vel, acc, jerk, goal, pe, re, ok, dt, gsi = torch.split(win, sizes, dim=-1)
        sq = (lambda x: x) if k > 1 else (lambda x: x[:, 0])  # noqa: E731
        lead = (P, k)
        return TrajOptResult(
            success=sq(ok[..., 0] > 0.5), knots=sq(kn.reshape(*lead, nk, D)), position=sq(pos.reshape(*lead, H, D)),
            position_error=sq(pe[..., 0]), rotation_error=sq(re[..., 0]), cost=sq(cost), seed_index=sq(idx),
            goal_config=sq(goal), traj_dt=sq(dt[..., 0]), velocity=sq(vel.reshape(*lead, H, D)),
            acceleration=sq(acc.reshape(*lead, H, D)), jerk=sq(jerk.reshape(*lead, H, D)), finetune_passes=passes,
            goalset_index=sq(gsi[..., 0].round().long()),
            all_seeds=dict(success=best["success"].view(P, S), traj_dt=best["dt"].view(P, S),
                           knots=best["knots"].view(P, S, nk, D), cost=ranked,
                           # why a seed failed: limits / collision over the optimiser's points, the same on the
                           # interpolated trajectory, pose error at the last point
                           feasible_rollout=best["feasible_rollout"].view(P, S), in_limits=best["in_limits"].view(P, S),
                           no_self_collision=best["no_self_collision"].view(P, S),
                           no_scene_collision=best["no_scene_collision"].view(P, S),
                           feasible_interpolated=best["feasible_interpolated"].view(P, S), converged=best["converged"].view(P, S),
                           position_error=best["pos_err"].view(P, S), rotation_error=best["rot_err"].view(P, S)))

    # ------------------------------------------------------------------ retiming
    def get_interpolated_trajectory(self, knots: torch.Tensor, start_position: torch.Tensor,
                                    goal_config: Optional[torch.Tensor] = None, retime: bool = True,
                                    traj_dt: Optional[torch.Tensor] = None):
        """Winner knots [P, n_knots, D] -> (position, velocity, acceleration, jerk) [P, steps, D] at
        ``cfg.interpolation_dt`` and the last valid step per trajectory (reference
        get_interpolated_trajectory, :579-634, BSPLINE_KNOTS_CUDA branch).  ``traj_dt`` [P]: the trajectories' dt
        (``TrajOptResult.traj_dt``; default ``cfg.rollout.traj_dt``); ``retime`` rescales it to the fastest one that
        respects the joint limits first (a solve has already done that for its winners)."""
        from ..backends import trajectory as trajectory_hip
        from ..util.trajectory import interpolate_bspline_knots

        rc, D = self.cfg.rollout, self.kin.num_dof
        P = knots.shape[0]
        dev = self.device
        sp = start_position.to(dev, torch.float32).reshape(-1, D)
        z = torch.zeros_like(sp)
        start = (sp, z, z, z)
        goal = None
        implicit = None
        if goal_config is not None:  # end at rest in the goal configuration (the optimiser's implicit goal state)
            zP = torch.zeros(P, D, device=dev)
            goal = (goal_config.to(dev, torch.float32).view(P, D), zP, zP, zP)
            implicit = torch.ones(P, dtype=torch.uint8, device=dev)
        traj_dt = (torch.full((P,), rc.traj_dt, device=dev) if traj_dt is None
                   else torch.as_tensor(traj_dt, dtype=torch.float32, device=dev).reshape(-1).expand(P).contiguous())
        if retime:
            H = rc.padded_horizon
            st = [torch.zeros(P, H, D, device=dev) for _ in range(4)]
            g = start if goal is None else goal
            sidx = (torch.zeros(P, dtype=torch.int32, device=dev) if sp.shape[0] == 1 else torch.arange(P, dtype=torch.int32, device=dev))
            gidx = torch.zeros(P, dtype=torch.int32, device=dev) if goal is None else torch.arange(P, dtype=torch.int32, device=dev)
            if goal is None and sp.shape[0] != 1:
                gidx = sidx
            imp = implicit if implicit is not None else torch.zeros(g[0].shape[0], dtype=torch.uint8, device=dev)
            dt_rows = traj_dt if goal is not None or sp.shape[0] != 1 else traj_dt[:1].contiguous()
            trajectory_hip.launch_bspline_interpolation_forward_kernel(
                *st, torch.zeros(P, device=dev), knots.contiguous(), *start, *g, sidx, gidx, dt_rows, imp, P, H, D, rc.n_knots,
                rc.bspline_degree)
            traj_dt = self.compute_trajectory_dt(st[1], st[2], st[3], traj_dt)
        knot_dt = traj_dt * rc.interpolation_steps
        out, last = interpolate_bspline_knots(knots, knot_dt, self.cfg.interpolation_dt, start, goal, implicit, rc.bspline_degree)
        return out, last, traj_dt


class _InterpolatedCheck:
    """Feasibility of optimised seeds on the trajectory re-sampled at the interpolation dt (reference
    ``_interpolate_and_compute_metrics`` + the ``interpolated_rollout`` constraints, solver_trajopt.py:475-497):
    single-dt B-spline re-interpolation -> FK -> self and (discrete) scene collision, joint limits on position /
    velocity / acceleration / jerk.  Samples after a trajectory's last step repeat its final state."""

    def __init__(self, kin: KinematicsParams, scene: Optional[SceneData], rc: TrajOptRolloutCfg):
        self.kin, self.scene, self.rc = kin, scene, rc
        self._shape = None

    def _alloc(self, B: int, n: int) -> None:
        if self._shape == (B, n):
            return
        k, d = self.kin, self.kin.device
        S, L, T = k.num_spheres, k.num_links, k.num_pose_links
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=d, dtype=dt)  # noqa: E731
        self.link_pos, self.link_quat, self.com = z(B, n, T, 3), z(B, n, T, 4), z(B, n, 4)
        self.spheres, self.cumul = z(B, n, S, 4), z(B, n, L, 3, 4)
        self.self_dist, self.self_grad, self.sparse = z(B, n, 1), z(B, n, S, 4), z(B, n, S, dt=torch.uint8)
        self.scene_dist, self.scene_grad = z(B, n, S), z(B, n, S, 4)
        self._pd, self._bbmv, self._bbmi = z(1), z(1), z(2, dt=torch.int16)
        self._one, self._zero = torch.ones(1, device=d), z(1)
        self._env0 = z(B, dt=torch.int32)
        self._shape = (B, n)

    def feasible(self, knots, dt, start, start_rows, goal, implicit, interpolation_dt, env_query_idx, static_steps=None) -> torch.Tensor:
        from ..backends import collision as collision_hip
        from ..backends import geometry as geometry_hip
        from ..backends import kinematics as kinematics_hip
        from ..util.trajectory import calculate_traj_steps, interpolate_bspline_knots

        k, rc, dev = self.kin, self.rc, self.kin.device
        B, nk, D = knots.shape
        knot_dt = dt * rc.interpolation_steps
        total = nk + rc.bspline_degree + 1
        if static_steps is not None:  # (a captured pass: the longest trajectory the dt range allows, no read-back)
            n = int(static_steps)
        else:
            _, steps_max = calculate_traj_steps(knot_dt, torch.full_like(knot_dt, float(interpolation_dt)), total + 1, nearest_int=True)
            n = -(-int(steps_max) // 32) * 32  # (one device -> host read per pass; buffers grow in steps of 32 samples)
        zs = torch.zeros_like(start)
        st = tuple(t[start_rows.long()].contiguous() for t in (start, zs, zs, zs))
        zg = torch.zeros_like(goal)
        gl = (goal.contiguous(), zg, zg, zg)
        imp = torch.full((B,), 1 if implicit else 0, dtype=torch.uint8, device=dev)
        (pos, vel, acc, jerk), last = interpolate_bspline_knots(knots, knot_dt, interpolation_dt, st, gl if implicit else None,
                                                                imp if implicit else None, rc.bspline_degree, out_steps=n,
                                                                out_steps_is_bound=static_steps is not None)
        self._alloc(B, n)
        S = k.num_spheres
        env = self._env0 if env_query_idx is None else env_query_idx
        kinematics_hip.launch_kinematics_forward_spheres(
            self.link_pos, self.link_quat, self.spheres, self.com, self.cumul, pos, k.fixed_transforms, k.link_spheres,
            k.link_masses_com, k.joint_map_type, k.joint_map, k.link_map, k.tool_frame_map, k.link_sphere_idx_map,
            k.joint_offset_map, env, k.num_envs, B * n, n, D, S, 32, True, False)
        sc = k.self_collision
        geometry_hip.self_collision_distance(
            self.self_dist, self.self_grad, self._pd, self.sparse, self.spheres, sc.sphere_padding, self._one,
            sc.collision_pairs, self._bbmv, self._bbmi, 1, 256, B, n, S, sc.collision_pairs.shape[0], False, True)
        ok = self.self_dist.view(B, -1).sum(-1) <= 0.0
        if self.scene is not None:
            collision_hip.sphere_obstacle_collision(
                self.scene_dist, self.scene_grad, self.spheres, self.scene.struct, self._one, self._zero, env, B, n, S,
                env_query_idx is not None, 0, False, None)
            ok &= self.scene_dist.view(B, -1).sum(-1) <= 0.0
        lo, hi = k.joint_limits_position[0], k.joint_limits_position[1]
        ok &= ((pos >= lo - 1e-4) & (pos <= hi + 1e-4)).all(-1).all(-1)
        vmax = k.joint_limits_velocity[1].abs()
        ok &= (vel.abs() <= vmax * (1.0 + 1e-3) + 1e-4).all(-1).all(-1)
        amax = joint_limit_vector(rc.max_acceleration, acc.shape[-1], acc.device)  # every joint against ITS OWN limit
        jmax = joint_limit_vector(rc.max_jerk, jerk.shape[-1], jerk.device)
        ok &= (acc.abs() <= amax * (1.0 + 1e-3) + 1e-4).all(-1).all(-1)
        ok &= (jerk.abs() <= jmax * (1.0 + 1e-3) + 1e-4).all(-1).all(-1)
        return ok
