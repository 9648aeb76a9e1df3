"""Collision-free trajectory optimisation to a goal pose: IK for the goal configuration, B-spline
seeds from the start to the IK solutions, L-BFGS on the full trajopt rollout of every seed in
parallel, best successful seed per problem.

Mirrors the flow of the reference ``TrajOptSolver._solve_impl`` (``curobo/_src/solver/
solver_trajopt.py:331-520``: goal IK -> linear seeds in joint space -> ``optimizer.optimize`` ->
metrics rollout -> success mask -> ranking, :469-484) for the ``lbfgs_bspline_trajopt.yml`` task.
"""

from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Optional

import torch

from ..distributed import global_argmin
from ..optim import LBFGSOpt, LBFGSOptCfg
from ..robot.kinematics_params import KinematicsParams
from ..rollout.trajopt_rollout import TrajOptRollout, TrajOptRolloutCfg
from ..scene.data import SceneData
from .ik import IKSolver, IKSolverCfg


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
    #: traj_dt 0.15 s: the reference optimises at its ``maximum_trajectory_dt`` and retimes afterwards
    rollout: TrajOptRolloutCfg = field(default_factory=lambda: TrajOptRolloutCfg(traj_dt=0.15))
    optimizer: LBFGSOptCfg = field(default_factory=lambda: LBFGSOptCfg(history=27, inner_iters=25, num_iters=100))
    ik: IKSolverCfg = field(default_factory=lambda: IKSolverCfg(num_seeds=32))
    seed: int = 0
    # retiming / re-interpolation of the winner (reference TrajOptSolverCfg: interpolation_dt,
    # minimum_trajectory_dt, maximum_trajectory_dt; solver_trajopt.py:579-680)
    interpolation_dt: float = 0.02
    minimum_trajectory_dt: float = 0.01
    maximum_trajectory_dt: float = 0.15


@dataclass
class TrajOptResult:
    success: torch.Tensor  # [P] bool
    knots: torch.Tensor  # [P, n_knots, D]
    position: torch.Tensor  # [P, H, D] interpolated joint trajectory of the winner
    position_error: torch.Tensor  # [P] at the last point
    rotation_error: torch.Tensor  # [P]
    cost: torch.Tensor  # [P]
    seed_index: torch.Tensor  # [P]
    goal_config: torch.Tensor  # [P, D] IK solution the seeds aim at
    ik_success: torch.Tensor  # [P]


class TrajOptSolver:
    def __init__(self, kin: KinematicsParams, scene: Optional[SceneData], num_problems: int,
                 cfg: Optional[TrajOptSolverCfg] = None, use_cuda_graph: bool = True):
        self.kin, self.scene, self.cfg = kin, scene, cfg or TrajOptSolverCfg()
        self.P, self.S, self.device = num_problems, self.cfg.num_seeds, kin.device
        # private copy of the optimiser configuration: the caller's cfg may be shared by solvers of other sizes
        ocfg = dataclasses.replace(self.cfg.optimizer, num_problems=self.P * self.S)
        self.cfg = dataclasses.replace(self.cfg, optimizer=ocfg)
        self.nls = len(ocfg.line_search_scale)
        rc = self.cfg.rollout
        self.ik = IKSolver(kin, scene, num_problems, self.cfg.ik, use_cuda_graph=use_cuda_graph)
        self.rollout = TrajOptRollout(kin, scene, self.P * self.S * self.nls, rc)
        self.metrics_rollout = TrajOptRollout(kin, scene, self.P * self.S, rc)
        self.K = max(1, min(self.cfg.num_ik_goals or self.S, self.S, self.cfg.ik.num_seeds))
        for r in (self.rollout, self.metrics_rollout):  # allocate the goal-state buffers before any graph capture
            r.update_goal_state(torch.zeros(self.P * self.K, kin.num_dof, device=self.device), None)
        bounds = (kin.joint_limits_position[0], kin.joint_limits_position[1])
        self.optimizer = LBFGSOpt(ocfg, self.rollout.cost_and_gradient, rc.n_knots, kin.num_dof, bounds, self.device,
                                  use_cuda_graph=use_cuda_graph)
        rows = torch.arange(self.P * self.S * self.nls, device=self.device)
        self._row_goal = (rows // (self.S * self.nls)).to(torch.int32)
        self._mrow_goal = (torch.arange(self.P * self.S, device=self.device) // self.S).to(torch.int32)

    def seed_goal_choice(self, ik_success: torch.Tensor) -> torch.Tensor:
        """[P, S] index (0..K-1) of the IK solution seed s of problem p ends in: s % K when that
        solution passed the IK checks, else the best one (solutions are ranked best first)."""
        s_goal = (torch.arange(self.S, device=self.device) % self.K).view(1, self.S).expand(self.P, self.S)
        ok = torch.gather(ik_success.view(self.P, self.K), 1, s_goal)
        return torch.where(ok, s_goal, torch.zeros_like(s_goal))

    def seed_knots(self, start: torch.Tensor, goal_config: torch.Tensor, choice: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[P, S, n_knots, D]: straight joint-space lines start -> the seed's goal configuration
        (reference seed generation, solver_trajopt.py:390-420; linspace weights as
        trajectory_seed_generator.py:150-170).  ``goal_config`` [P, K, D] (or [P, D]), ``choice``
        [P, S] from ``seed_goal_choice``; a seed that repeats an earlier seed's goal adds a smooth
        mid-trajectory bump so that the seeds stay distinct."""
        rc, D, P, S = self.cfg.rollout, self.kin.num_dof, self.P, self.S
        goal_config = goal_config.reshape(P, -1, D)
        if choice is None:
            choice = torch.zeros(P, S, dtype=torch.int64, device=self.device)
        goal = torch.gather(goal_config, 1, choice.unsqueeze(-1).expand(P, S, D))  # [P, S, D]
        t = torch.linspace(0.0, 1.0, rc.n_knots + 2, device=self.device)[1:-1].view(1, 1, -1, 1)
        line = start.view(1, 1, 1, D) * (1 - t) + goal.view(P, S, 1, D) * t
        gen = torch.Generator(device="cpu").manual_seed(self.cfg.seed)
        half = 0.5 * (self.kin.joint_limits_position[1] - self.kin.joint_limits_position[0])
        bump = torch.randn(P, S, 1, D, generator=gen).to(self.device) * self.cfg.seed_bump * half
        sidx = torch.arange(S, device=self.device).view(1, S)
        first_use = (choice == sidx) & (sidx < goal_config.shape[1])  # seed s is the first one aimed at goal s
        bump = bump * (~first_use).view(P, S, 1, 1)
        knots = line + bump * torch.sin(torch.pi * t)
        lo, hi = self.kin.joint_limits_position[0], self.kin.joint_limits_position[1]
        return torch.minimum(torch.maximum(knots, lo + 1e-3), hi - 1e-3).contiguous()

    def solve_pose(self, start_position: torch.Tensor, goal_position: torch.Tensor, goal_quat: torch.Tensor,
                   env_idx: Optional[torch.Tensor] = None) -> TrajOptResult:
        """One shared start configuration [D]; goal_position [P, 3], goal_quat [P, 4] (wxyz); ``env_idx``
        [P]: problem p plans in scene environment env_idx[p] (reference batch-env planning,
        motion_planner_batch.py; ``idxs_env`` / ``use_multi_env`` of the collision costs)."""
        P, S, D, T = self.P, self.S, self.kin.num_dof, self.kin.num_pose_links
        rc = self.cfg.rollout
        start = start_position.to(self.device, torch.float32).view(1, D)
        K = self.K
        # the L-BFGS stage always runs here: the goal configurations should be converged, not just inside the IK
        # tolerances (the reference's motion planner switches exit_early off as well, motion_planner.py:143-144)
        ikr = self.ik.solve_pose(goal_position, goal_quat, return_seeds=K, exit_early=False, env_idx=env_idx)
        mode = env_idx is not None
        if mode != getattr(self, "_env_mode", False):
            self.optimizer._graph = None  # the multi-env flag is a kernel argument: capture again
        self._env_mode = mode
        env = env_idx.to(self.device).long().view(P) if mode else None
        ik_ok = ikr.success.view(P, K)
        ik_q = ikr.solution.reshape(P, K, D).contiguous()
        choice = self.seed_goal_choice(ik_ok)  # [P, S]
        goal_row = torch.arange(P, device=self.device).view(P, 1) * K + choice  # row of ik_q.view(P*K, D) per (p, s)
        gp = goal_position.to(self.device, torch.float32).view(P, 1, 1, 3).expand(P, T, 1, 3).contiguous()
        gq = goal_quat.to(self.device, torch.float32).view(P, 1, 1, 4).expand(P, T, 1, 4).contiguous()
        grows = (goal_row.view(P, S, 1).expand(P, S, self.nls).reshape(-1), goal_row.reshape(-1))
        for r, rows, gr in ((self.rollout, self._row_goal, grows[0]), (self.metrics_rollout, self._mrow_goal, grows[1])):
            r.update_start_state(start)
            r.update_env_query_idx(env[rows.long()] if mode else None)
            r.update_goals(gp, gq, rows)
            r.update_goal_state(ik_q.view(P * K, D), gr)  # end at rest in the seed's IK solution (implicit goal state)
        seeds = self.seed_knots(start, ik_q, choice)
        best = self.optimizer.optimize(seeds.view(P * S, rc.n_knots, D))
        knots = best.reshape(P * S, rc.n_knots * D).contiguous()
        m = self.metrics_rollout
        cost = m.evaluate_action(knots.view(P * S, rc.n_knots, D), with_gradient=False)
        pos_err = m.pose_pos_dist.view(P, S, -1, T)[:, :, -1, 0]
        rot_err = m.pose_rot_dist.view(P, S, -1, T)[:, :, -1, 0]
        lo, hi = self.kin.joint_limits_position[0], self.kin.joint_limits_position[1]
        q = m.position.view(P, S, -1, D)
        feasible = ((q >= lo - 1e-4) & (q <= hi + 1e-4)).all(-1).all(-1)
        # velocity / acceleration / jerk inside their limits at the optimised dt: the reference's success mask is the
        # c-space STATE constraint at zero activation distance over position, velocity, acceleration, jerk (and torque)
        # (content/configs/task/metrics_base.yml:16-19, solver/solver_trajopt_result.py:154-210)
        for x, b in ((m.velocity, m._v_b), (m.acceleration, m._a_b), (m.jerk, m._j_b)):
            x = x.view(P, S, -1, D)
            feasible &= ((x >= b[0] - 1e-3 * b[0].abs() - 1e-4) & (x <= b[1] + 1e-3 * b[1].abs() + 1e-4)).all(-1).all(-1)
        feasible &= m.self_dist.view(P, S, -1).sum(-1) <= 0.0
        if self.scene is not None:
            feasible &= m.scene_dist.view(P, S, -1).sum(-1) <= 0.0
        if rc.use_torque_limits:  # inverse-dynamics torques of the whole trajectory inside the effort limits
            feasible &= (m._tau.view(P, S, -1, D).abs() <= m._effort_b[1] * (1.0 + 1e-3) + 1e-3).all(-1).all(-1)
        ok = feasible & (pos_err < self.cfg.position_threshold) & (rot_err < self.cfg.rotation_threshold)
        ranked = cost.view(P, S) + 1e16 * (~ok).float()  # reference solver_trajopt.py:469-484
        payload = torch.cat([knots.view(P, S, -1), pos_err.unsqueeze(-1), rot_err.unsqueeze(-1), ok.float().unsqueeze(-1),
                             cost.view(P, S, 1)], dim=-1)
        _, idx, win = global_argmin(ranked, payload, 0)
        V = rc.n_knots * D
        ar = torch.arange(P, device=self.device)
        widx = idx.clamp(0, S - 1)
        traj = q[ar, widx]
        win_goal = ik_q[ar, choice[ar, widx]]  # the IK solution the winning seed ends in
        return TrajOptResult(success=win[:, V + 2] > 0.5, knots=win[:, :V].view(P, rc.n_knots, D), position=traj,
                             position_error=win[:, V], rotation_error=win[:, V + 1], cost=win[:, V + 3], seed_index=idx,
                             goal_config=win_goal, ik_success=ik_ok[:, 0])

    # ------------------------------------------------------------------ retiming
    def compute_trajectory_dt(self, velocity: torch.Tensor, acceleration: torch.Tensor, jerk: torch.Tensor,
                              dt: Optional[torch.Tensor] = None, epsilon: float = 1e-3) -> torch.Tensor:
        """reference TrajOptSolver.compute_trajectory_dt (:636-677): the dt at which the trajectory just
        respects the velocity / acceleration / jerk limits, clamped to [minimum, maximum]_trajectory_dt;
        inputs [B, H, D] sampled at ``dt`` [B] (or scalar ``cfg.rollout.traj_dt``)."""
        from ..util.trajectory import calculate_dt_no_clamp

        rc, D = self.cfg.rollout, self.kin.num_dof
        ones = torch.ones(D, device=self.device)
        vmax = self.kin.joint_limits_velocity[1].abs()
        score = calculate_dt_no_clamp(velocity, acceleration, jerk, vmax, rc.max_acceleration * ones, rc.max_jerk * ones, epsilon)
        base = dt if dt is not None else torch.full_like(score, rc.traj_dt)
        return torch.clamp(score * base, min=self.cfg.minimum_trajectory_dt, max=self.cfg.maximum_trajectory_dt)

    def get_interpolated_trajectory(self, knots: torch.Tensor, start_position: torch.Tensor,
                                    goal_config: Optional[torch.Tensor] = None, retime: bool = True):
        """Winner knots [P, n_knots, D] -> (position, velocity, acceleration, jerk) [P, steps, D] at
        ``cfg.interpolation_dt`` and the last valid step per trajectory (reference
        get_interpolated_trajectory, :579-634, BSPLINE_KNOTS_CUDA branch), after rescaling the
        trajectory's dt to the fastest one that respects the joint limits when ``retime``."""
        from ..backends import trajectory as trajectory_hip
        from ..util.trajectory import interpolate_bspline_knots

        rc, D = self.cfg.rollout, self.kin.num_dof
        P = knots.shape[0]
        dev = self.device
        z = torch.zeros(1, D, device=dev)
        start = (start_position.to(dev, torch.float32).view(1, D), z, z, z)
        goal = None
        implicit = None
        if goal_config is not None:  # end at rest in the goal configuration (the optimiser's implicit goal state)
            zP = torch.zeros(P, D, device=dev)
            goal = (goal_config.to(dev, torch.float32).view(P, D), zP, zP, zP)
            implicit = torch.ones(P, dtype=torch.uint8, device=dev)
        traj_dt = torch.full((P,), rc.traj_dt, device=dev)
        if retime:
            H = rc.padded_horizon
            st = [torch.zeros(P, H, D, device=dev) for _ in range(4)]
            g = start if goal is None else goal
            sidx = torch.zeros(P, dtype=torch.int32, device=dev)
            gidx = sidx if goal is None else torch.arange(P, dtype=torch.int32, device=dev)
            imp = implicit if implicit is not None else torch.zeros(1, dtype=torch.uint8, device=dev)
            trajectory_hip.launch_bspline_interpolation_forward_kernel(
                *st, torch.zeros(P, device=dev), knots.contiguous(), *start, *g, sidx, gidx, traj_dt, imp, P, H, D, rc.n_knots,
                rc.bspline_degree)
            traj_dt = self.compute_trajectory_dt(st[1], st[2], st[3], traj_dt)
        knot_dt = traj_dt * rc.interpolation_steps
        out, last = interpolate_bspline_knots(knots, knot_dt, self.cfg.interpolation_dt, start, goal, implicit, rc.bspline_degree)
        return out, last, traj_dt
