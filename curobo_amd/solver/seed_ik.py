"""Levenberg-Marquardt seed-IK solver on the HIP kernels.

Mirrors the reference's ``SeedIKSolver`` (``curobo/_src/solver/seed_ik/seed_ik_solver.py:48-824``,
``seed_ik_error_calculator.py``, ``seed_iteration_state_manager.py``, ``seed_ik_solver_cfg.py``):
fast approximate IK solutions from Halton seeds that seed the L-BFGS IK solver.  One LM iteration
is five launches here (the reference: a Warp tile kernel, two CUDA kernels, a Warp cost kernel and
~25 torch elementwise kernels under a CUDA graph):

    lm_step (J^T J on the matrix cores + Cholesky solve)   csrc/linalg.hip
    FK + geometric tool Jacobian of the candidate          csrc/kinematics.hip
    tool-pose cost, its position / quaternion gradients    csrc/cost.hip
    FK VJP of those gradients = pose J^T e                 csrc/kinematics.hip
    joint-limit rows + trust ratio + accept / damping /
    selection / convergence flags                          csrc/seed_ik.hip

``inner_iterations`` iterations are captured into one hipGraph; the early-exit test between
replays is the reference's (`_calculate_exit_condition`).  ``solve_batch(current_position=, dt=)``
tightens the joint-limit bounds to what one step of ``dt`` can reach from the current position
(velocity clamping, seed_ik_error_calculator.py:355-363, seed_ik_solver.py:601-615).  The velocity /
acceleration residual rows (``cfg.velocity_weight``, ``cfg.acceleration_weight``, default 0 as in the reference;
seed_ik_error_calculator.py:389-456) are diagonal blocks like the joint-limit rows and are folded into them in the
state-update kernel (``solve_batch(..., current_velocity=)`` for the acceleration rows).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import math

import torch

from ..backends import cost as cost_hip
from ..backends import kinematics as kinematics_hip
from ..backends import linalg as linalg_hip
from ..robot.kinematics_params import KinematicsParams


@dataclass
class SeedIKSolverCfg:
    """Names and defaults of ``solver/seed_ik/seed_ik_solver_cfg.py:25-94``."""

    max_iterations: int = 16
    inner_iterations: int = 4
    position_tolerance: float = 0.005
    orientation_tolerance: float = 0.05
    convergence_position_tolerance: float = 0.00001
    convergence_orientation_tolerance: float = 0.00001
    convergence_joint_limit_weight: float = 1.0
    lambda_initial: float = 0.2
    lambda_factor: float = 2.0
    lambda_max: float = 1.0e10
    lambda_min: float = 1e-5
    joint_limit_margin: float = 0.001
    batch_success_threshold: float = 1.0
    num_seeds: int = 1
    joint_limit_weight: float = 1.0
    use_cuda_graph: bool = True
    rho_min: float = 1e-3
    sampler_seed: int = 451
    start_cspace_dist_weight: float = 0.01
    position_weight: float = 1.0
    orientation_weight: float = 1.0
    #: velocity / acceleration regularisation rows (reference SeedIKSolverCfg.velocity_weight / acceleration_weight,
    #: seed_ik_error_calculator.py:389-456): active with ``solve_batch(current_position=, dt=[, current_velocity=])``
    velocity_weight: float = 0.0
    acceleration_weight: float = 0.0
    #: every inner block of LM iterations (and the initial evaluation) as ONE launch with the per-problem state in LDS
    #: (``curobo_hip_seed_ik_iterate``) instead of five launches per iteration; same per-stage arithmetic except
    #: J^T J (fp32 FMAs on the row instead of the matrix-core contraction of the stand-alone LM step)
    fused_iterations: bool = True

    def __post_init__(self):
        if self.max_iterations < self.inner_iterations or self.max_iterations % self.inner_iterations != 0:
            raise ValueError(f"max_iterations: {self.max_iterations} must be a positive multiple of inner_iterations: "
                             f"{self.inner_iterations}")


@dataclass
class SeedIKResult:  # the fields of IKSolverResult the seed solver fills (solver_ik_result.py)
    success: torch.Tensor         # [P, return_seeds] bool
    solution: torch.Tensor        # [P, return_seeds, D]
    position_error: torch.Tensor  # [P, return_seeds]
    rotation_error: torch.Tensor  # [P, return_seeds]
    #: LM iterations that ran: an int, or (the one-graph solve) a device counter of the blocks of iterations that ran,
    #: read -- one device -> host round trip -- only when somebody asks (``iterations``)
    iterations_or_counter: object = 0
    inner_iterations: int = 1

    @property
    def iterations(self) -> int:
        v = self.iterations_or_counter
        return int(v.item()) * self.inner_iterations if torch.is_tensor(v) else int(v)


def _packed_outputs(device, specs):
    """tensors of the given (shape, dtype) list as views of ONE byte buffer (16-byte aligned slices): (buffer, views)"""
    import math as _m

    sizes = [int(_m.prod(sh)) * torch.empty((), dtype=dt).element_size() for sh, dt in specs]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += (n + 15) & ~15
    buf = torch.zeros(max(o, 16), dtype=torch.uint8, device=device)
    return buf, _views_of(buf, specs, offs, sizes)


def _views_of(buf, specs, offs, sizes):
    return [buf[o:o + n].view(dt).view(*sh) for (sh, dt), o, n in zip(specs, offs, sizes)]


def _unpack_like(views, buf_copy):
    """the views of ``_packed_outputs`` (same shapes, dtypes, order) re-made over a copy of its buffer"""
    specs = [(tuple(v.shape), v.dtype) for v in views]
    sizes = [v.numel() * v.element_size() for v in views]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += (n + 15) & ~15
    return _views_of(buf_copy, specs, offs, sizes)


class HaltonSeeds:
    """The reference's ``SampleBuffer.create_halton_sample_buffer`` (``util/sampling/sample_buffer.py``
    :60-157,253-279 over ``sequencer_halton.py``: scipy's scrambled Halton sequence): 2000 points are
    generated once, then ``get_samples(n)`` draws n of them by random index.  The index stream comes
    from a CPU ``torch.Generator`` here (the reference's lives on its device, so it is device specific)."""

    def __init__(self, ndims: int, low: torch.Tensor, high: torch.Tensor, seed: int = 123, store_buffer: int = 2000):
        from scipy.stats.qmc import Halton

        self.low, self.range = low, high - low
        self._qmc_args = (ndims, seed)
        self.buffer = torch.as_tensor(Halton(d=ndims, seed=seed, scramble=True).random(store_buffer), dtype=torch.float32,
                                      device=low.device)
        self._gen = torch.Generator(device="cpu").manual_seed(seed)
        self._state0 = self._gen.get_state().clone()

    def reset(self) -> None:
        self._gen.set_state(self._state0)

    def get_samples(self, n: int, bounded: bool = True) -> torch.Tensor:
        idx = torch.randint(0, self.buffer.shape[0], (n,), generator=self._gen).to(self.buffer.device)
        s = self.buffer[idx]
        return s * self.range + self.low if bounded else s

    def get_samples_host(self, n: int) -> torch.Tensor:
        """the same n bounded samples as ``get_samples(n)`` (same index stream, the same two fp32 roundings), computed on
        the host: a solve then uploads ONE small row instead of running an index / gather / multiply / add chain of
        launches for it"""
        if getattr(self, "_host", None) is None:
            self._host = (self.buffer.cpu(), self.range.cpu(), self.low.cpu())
        buf, rng, low = self._host
        idx = torch.randint(0, buf.shape[0], (n,), generator=self._gen)
        return buf[idx] * rng + low


class SeedIKSolver:
    """``solve_batch(goal_position[P, T, 3], goal_quat[P, T, 4] (wxyz))`` -> best ``return_seeds``
    configurations per problem out of ``num_seeds`` LM runs."""

    def __init__(self, kin: KinematicsParams, num_problems: int, cfg: Optional[SeedIKSolverCfg] = None,
                 default_joint_position: Optional[torch.Tensor] = None, num_goalset: int = 1,
                 seed_offset: int = 0, global_num_seeds: Optional[int] = None):
        """``num_goalset`` G > 1: every problem has G alternative goal poses per tool frame and each seed
        is pulled to the member its pose error is smallest for (reference SeedIKSolver goal-set
        buffer, seed_ik_solver.py:340-365; the arg-min is the pose kernel's).

        Seed shards (one process per GPU): this instance runs seeds
        ``[seed_offset, seed_offset + cfg.num_seeds)`` of a set of ``global_num_seeds`` that every rank
        generates identically (same sampler seed, same index stream) and slices, so any world size
        works on the same seeds; ``solve_batch`` then ranks the runs of ALL ranks (one all-gather)."""
        self.kin, self.cfg = kin, cfg or SeedIKSolverCfg()
        self.G = num_goalset
        c, dev = self.cfg, kin.fixed_transforms.device
        self.device = dev
        self.P, self.S = num_problems, c.num_seeds
        self.seed_offset = int(seed_offset)
        self.S_global = int(global_num_seeds) if global_num_seeds is not None else self.S
        if not (0 <= self.seed_offset and self.seed_offset + self.S <= self.S_global):
            raise ValueError(f"seed shard [{self.seed_offset}, {self.seed_offset + self.S}) is not inside the global seed "
                             f"set of {self.S_global}")
        self.n = n = self.P * self.S
        D, T, L, Sp = kin.num_dof, kin.num_pose_links, kin.num_links, kin.num_spheres
        self.D, self.T, self.R = D, T, 6 * T + D
        lo, hi = kin.joint_limits_position[0], kin.joint_limits_position[1]
        margin = (hi - lo) * c.joint_limit_margin
        self.action_min, self.action_max = (lo + margin).contiguous(), (hi - margin).contiguous()
        self._limits = (lo, hi)
        self.default_joint_position = (default_joint_position if default_joint_position is not None
                                       else 0.5 * (lo + hi)).to(dev, torch.float32)
        self.sampler = HaltonSeeds(D, self.action_min, self.action_max, seed=c.sampler_seed)
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)  # noqa: E731
        # iteration state (reference SeedIKState)
        self.q, self.jacobian, self.jTerror = z(n, D), z(n, self.R, D), z(n, D)
        self.error_norm, self.position_error, self.orientation_error = z(n), z(n), z(n)
        self.lambda_damping = z(n)
        self.success, self.improvement = z(n, dt=torch.uint8), z(n, dt=torch.uint8)
        # candidate evaluation
        self.q_new, self.pred_reduction = z(n, D), z(n)
        self.link_pos, self.link_quat = z(n, T, 3), z(n, T, 4)
        self.pose_jacobian, self.cumul_mat = z(n, T, 6, D), z(n, L, 3, 4)
        self.robot_spheres, self.com = z(n, max(Sp, 1), 4), z(n, 4)
        self.pose_cost, self.pos_dist, self.rot_dist = z(n, T, 2), z(n, T), z(n, T)
        self.grad_pos, self.grad_quat = z(n, T, 3), z(n, T, 4)
        self.goalset_idx = z(n, T, dt=torch.int32)
        self.pose_jTerror = z(n, D)
        self.env_query_idx = z(n, dt=torch.int32)
        self.idxs_goal = (torch.arange(n, device=dev) // self.S).to(torch.int32)
        self.goal_position, self.goal_quat = z(self.P, T, self.G, 3), z(self.P, T, self.G, 4)
        self.goal_quat[..., 0] = 1.0
        self._pose_w = torch.tensor([c.position_weight, c.orientation_weight], device=dev)
        self._axes_w = torch.ones(T * 6, device=dev)
        self._tol = z(T * 2)
        self._project = z(T, dt=torch.uint8)
        # velocity clamping of the bounds: buffers with stable addresses (captured graphs), one graph per mode
        self._vel_current, self._vel_dt = z(n, D), torch.ones(n, device=dev)
        self._vel_limits = kin.joint_limits_velocity.to(dev, torch.float32).contiguous()
        self._vel_velocity = z(n, D)  # current joint velocity (acceleration rows)
        self._vel_active = False
        self._graphs = {}
        self._stop_flag = z(1, dt=torch.int32)
        self._solve_graphs, self._last_outer = {}, 0
        self._blocks_run = z(1, dt=torch.int32)
        self._fused_fits: Optional[bool] = None

    # ------------------------------------------------------------------ one evaluation / iteration
    def _evaluate_candidate(self, q: torch.Tensor, initial: bool) -> None:
        """reference SeedIKErrorCalculator.compute_error_and_jacobian + (unless ``initial``)
        SeedIterationStateManager.update_iteration_state"""
        k, c, n, D, T = self.kin, self.cfg, self.n, self.D, self.T
        kinematics_hip.launch_kinematics_forward_spheres_jacobian(
            self.link_pos, self.link_quat, self.robot_spheres, self.com, self.pose_jacobian, self.cumul_mat, q,
            k.fixed_transforms, k.link_spheres, k.link_masses_com, k.joint_map_type, k.joint_map, k.link_map,
            k.tool_frame_map, k.link_sphere_idx_map, k.link_chain_data, k.link_chain_offsets, k.joint_links_data,
            k.joint_links_offsets, k.joint_affects_endeffector, k.joint_offset_map, self.env_query_idx, k.num_envs,
            n, 1, D, k.num_spheres, 32, True, False)
        cost_hip.tool_pose_distance(
            self.pose_cost, self.pos_dist, self.rot_dist, self.grad_pos, self.grad_quat, self.goalset_idx, self.link_pos,
            self.link_quat, self.goal_position, self.goal_quat, self.idxs_goal, self._pose_w, self._axes_w, self._axes_w,
            self._tol, self._tol, self._project, n, 1, T, self.G, 0)
        kinematics_hip.launch_kinematics_backward(
            self.pose_jTerror, self.grad_pos, self.grad_quat, self.robot_spheres, self.com, self.com, self.grad_pos,
            self.cumul_mat, k.link_spheres, k.link_masses_com, k.link_map, k.joint_map, k.joint_map_type,
            k.tool_frame_map, k.link_sphere_idx_map, k.link_chain_data, k.link_chain_offsets, k.joint_links_data,
            k.joint_links_offsets, k.joint_affects_endeffector, k.joint_offset_map, self.env_query_idx, k.num_envs,
            n, 1, D, 0, False, False)
        linalg_hip.seed_ik_update_state(
            self.q, self.jacobian, self.jTerror, self.error_norm, self.position_error, self.orientation_error,
            self.lambda_damping, self.success, self.improvement, q, self.pose_jacobian.view(n, 6 * T, D),
            self.pose_jTerror, self.pose_cost, self.pos_dist, self.rot_dist, self.pred_reduction, self.action_min,
            self.action_max, *((self._vel_current, self._vel_dt, self._vel_limits) if self._vel_active else (None, None, None)),
            c.joint_limit_weight, c.rho_min, c.lambda_factor, c.lambda_min,
            c.lambda_max, c.convergence_position_tolerance, c.convergence_orientation_tolerance,
            c.convergence_joint_limit_weight, initial,
            current_velocity=self._vel_velocity if (self._vel_active and c.acceleration_weight > 0) else None,
            velocity_weight=c.velocity_weight if self._vel_active else 0.0,
            acceleration_weight=c.acceleration_weight if self._vel_active else 0.0)

    def _fused_ok(self) -> bool:
        """the one-launch iteration needs dof <= 16 and 16 problems' state in 64 KB of LDS (a 7-dof arm with more than
        ~27 links, or a dual arm with two tool frames, does not fit): otherwise the five-launch iteration runs"""
        if self._fused_fits is None:
            k = self.kin
            self._fused_fits = bool(self.cfg.fused_iterations) and self.D <= 16 and linalg_hip.seed_ik_iterate_fits(
                self.D, k.num_links, self.T, int(k.link_chain_data.shape[0]))
        return self._fused_fits

    def _iterate_fused(self, iterations: int, seeds: Optional[torch.Tensor] = None) -> None:
        """``iterations`` LM iterations (after the initial evaluation of ``seeds`` when given) in one launch"""
        k, c = self.kin, self.cfg
        vel = self._vel_active
        linalg_hip.seed_ik_iterate(
            self.q, self.jacobian, self.jTerror, self.error_norm, self.position_error, self.orientation_error,
            self.lambda_damping, self.success, self.improvement, seeds, self.goal_position, self.goal_quat, self.idxs_goal,
            self._pose_w, self._axes_w, self._tol, self._project, self.G, 0, k.fixed_transforms, k.joint_map_type, k.joint_map,
            k.link_map, k.tool_frame_map, k.link_chain_data, k.link_chain_offsets, k.joint_links_data, k.joint_links_offsets,
            k.joint_affects_endeffector, k.joint_offset_map, self.action_min, self.action_max,
            *((self._vel_current, self._vel_dt, self._vel_limits) if vel else (None, None, None)),
            c.joint_limit_weight, c.rho_min, c.lambda_factor, c.lambda_min, c.lambda_max, c.convergence_position_tolerance,
            c.convergence_orientation_tolerance, c.convergence_joint_limit_weight, iterations, seeds is not None,
            current_velocity=self._vel_velocity if (vel and c.acceleration_weight > 0) else None,
            velocity_weight=c.velocity_weight if vel else 0.0, acceleration_weight=c.acceleration_weight if vel else 0.0,
            stop_flag=None if seeds is not None else self._stop_flag, blocks_run=None if seeds is not None else self._blocks_run)

    def _lm_iteration(self) -> None:
        linalg_hip.levenberg_marquardt_step(self.q_new, self.pred_reduction, self.jacobian, self.jTerror,
                                            self.lambda_damping, self.q)
        self._evaluate_candidate(self.q_new, initial=False)

    def _inner_iterations(self) -> None:
        for _ in range(self.cfg.inner_iterations):
            self._lm_iteration()

    def _run_inner(self) -> None:
        if not self.cfg.use_cuda_graph:
            self._inner_iterations()
            return
        if self._vel_active not in self._graphs:
            saved = [t.clone() for t in self._state()]
            self._lm_iteration()  # warm-up outside the capture
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._inner_iterations()
            for t, s in zip(self._state(), saved):
                t.copy_(s)
            self._graphs[self._vel_active] = graph
        self._graphs[self._vel_active].replay()

    def _state(self):
        return [self.q, self.jacobian, self.jTerror, self.error_norm, self.position_error, self.orientation_error,
                self.lambda_damping, self.success, self.improvement]

    # ------------------------------------------------------------------ seeds, solve
    def generate_seeds(self, seed_config: Optional[torch.Tensor] = None) -> torch.Tensor:
        """reference _generate_seed_configs (:470-520): the same Halton seeds for every problem, the
        last one replaced by the default joint position; given seeds come first.  A seed shard builds the
        global set and keeps its slice (the index stream is a host generator: every rank draws the same)."""
        P, S, D, SG, lo = self.P, self.S, self.D, self.S_global, self.seed_offset
        if seed_config is not None:
            seed_config = seed_config.to(self.device, torch.float32).view(P, -1, D)
            if seed_config.shape[1] > SG:
                raise ValueError(f"seed_config has {seed_config.shape[1]} seeds, but only {SG} are needed")
            if seed_config.shape[1] == SG:
                return seed_config[:, lo:lo + S].contiguous()
            extra = self.sampler.get_samples(P * (SG - seed_config.shape[1])).view(P, -1, D)
            return torch.cat([seed_config, extra], dim=1)[:, lo:lo + S].contiguous()
        seeds = self.sampler.get_samples(SG).view(1, SG, D).repeat(P, 1, 1)
        seeds[:, -1, :] = self.default_joint_position.view(1, -1)
        return seeds[:, lo:lo + S].contiguous()

    def _default_seed_row(self) -> torch.Tensor:
        """the seed set of ONE problem when no seeds are given (``generate_seeds(None)`` repeats it for every problem): host
        tensor [S_global, D], the same values from the same index stream"""
        row = self.sampler.get_samples_host(self.S_global)
        if getattr(self, "_default_host", None) is None:
            self._default_host = self.default_joint_position.cpu()
        row[-1] = self._default_host
        return row

    def reset_seed(self) -> None:
        self.sampler.reset()

    def _sharded(self) -> bool:
        import torch.distributed as dist

        return self.S_global != self.S and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def solve_batch(self, goal_position: torch.Tensor, goal_quat: torch.Tensor, seed_config: Optional[torch.Tensor] = None,
                    return_seeds: int = 1, current_position: Optional[torch.Tensor] = None,
                    dt: Optional[torch.Tensor] = None, current_velocity: Optional[torch.Tensor] = None) -> SeedIKResult:
        """``current_position`` [P, D]: first seed of every problem and the c-space distance term of the
        ranking; together with ``dt`` (scalar or [P]) it also switches velocity clamping on: a solution
        has to lie within ``velocity_limits * dt`` of the current position (reference: ``current_state.dt``)."""
        P, S, D, T, c = self.P, self.S, self.D, self.T, self.cfg
        self._vel_active = current_position is not None and dt is not None
        if self._vel_active:
            self._vel_current.view(P, S, D).copy_(current_position.to(self.device, torch.float32).view(P, 1, D).expand(P, S, D))
            self._vel_dt.view(P, S).copy_(torch.as_tensor(dt, dtype=torch.float32, device=self.device).reshape(-1, 1).expand(P, S))
            if current_velocity is not None:  # acceleration rows (cfg.acceleration_weight): [P, D]
                self._vel_velocity.view(P, S, D).copy_(current_velocity.to(self.device, torch.float32).view(P, 1, D).expand(P, S, D))
            else:
                self._vel_velocity.zero_()
        if goal_position is not self.goal_position:  # (IKSolver shares its goal buffers with this solver: already in place)
            self.goal_position.copy_(goal_position.to(self.device, torch.float32).view(P, T, self.G, 3))
            self.goal_quat.copy_(goal_quat.to(self.device, torch.float32).view(P, T, self.G, 4))
        if seed_config is None and current_position is not None:
            seed_config = current_position.view(P, 1, D)
        fused = self._fused_ok()
        one_graph = fused and c.use_cuda_graph and not self._vel_active and current_position is None and not self._sharded()
        if one_graph and seed_config is None and getattr(self, "_seeds_static", None) is not None:
            # the default seed set is one row of [S, D] repeated for every problem: built on the host (same values), one
            # small upload and one broadcast copy into the graph's seed buffer instead of ~8 launches
            lo = self.seed_offset
            row = self._default_seed_row()[lo:lo + S].contiguous()
            if getattr(self, "_seed_row_dev", None) is None:
                self._seed_row_dev = torch.empty(S, D, device=self.device)
                self._seed_row_pin = torch.empty(S, D).pin_memory() if self.device.type == "cuda" else torch.empty(S, D)
                self._seed_row_done = torch.cuda.Event() if self.device.type == "cuda" else None
                self._seed_row_pending = False
            # the upload of the previous solve may still be queued (solve_batch does not synchronise): the pinned buffer is
            # rewritten only after the stream has read it, else that solve would run on this call's seed row
            if self._seed_row_pending:
                self._seed_row_done.synchronize()
            self._seed_row_pin.copy_(row)
            self._seed_row_dev.copy_(self._seed_row_pin, non_blocking=True)
            if self._seed_row_done is not None:
                self._seed_row_done.record()
                self._seed_row_pending = True
            self._seeds_static.view(P, S, D).copy_(self._seed_row_dev.view(1, S, D).expand(P, S, D))
            seeds = None
        else:
            seeds = self.generate_seeds(seed_config).reshape(self.n, D).contiguous()
        if one_graph:
            # the whole solve after the seeds (initial evaluation, every block of iterations with the device-side exit
            # test, the ranking) replayed from ONE hipGraph: ~35 small submissions become one
            if return_seeds not in self._solve_graphs:
                if getattr(self, "_seeds_static", None) is None:
                    self._seeds_static = torch.empty_like(seeds)
                if seeds is not None:  # (None: the broadcast above already filled the buffer)
                    self._seeds_static.copy_(seeds)
                self._solve_from_seeds(self._seeds_static, return_seeds, None)  # warm-up outside the capture
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out = self._solve_from_seeds(self._seeds_static, return_seeds, None)
                self._solve_graphs[return_seeds] = (graph, out, self._select_pack)
            elif seeds is not None:
                self._seeds_static.copy_(seeds)
            graph, out, pack = self._solve_graphs[return_seeds]
            graph.replay()
            # the four outputs live in ONE buffer (see _solve_from_seeds): one copy instead of four
            ok_t, sol, pos_t, ori_t = _unpack_like(out, pack.clone())
            return SeedIKResult(success=ok_t, solution=sol, position_error=pos_t, rotation_error=ori_t,
                                iterations_or_counter=self._blocks_run, inner_iterations=c.inner_iterations)
        ok_t, sol, pos_t, ori_t = self._solve_from_seeds(seeds, return_seeds, current_position)
        if fused:
            return SeedIKResult(success=ok_t, solution=sol, position_error=pos_t, rotation_error=ori_t,
                                iterations_or_counter=self._blocks_run, inner_iterations=c.inner_iterations)
        return SeedIKResult(success=ok_t, solution=sol, position_error=pos_t, rotation_error=ori_t,
                            iterations_or_counter=(self._last_outer + 1) * c.inner_iterations)

    def _solve_from_seeds(self, seeds: torch.Tensor, return_seeds: int, current_position: Optional[torch.Tensor]):
        """initial evaluation -> blocks of LM iterations (exit test between blocks) -> ranked top ``return_seeds``"""
        P, S, D, c = self.P, self.S, self.D, self.cfg
        self.lambda_damping.fill_(c.lambda_initial)
        self.success.zero_()
        fused = self._fused_ok()
        outer = c.max_iterations // c.inner_iterations
        if fused:
            # every block of iterations is enqueued at once: the exit test between blocks runs on the device and the
            # remaining launches return immediately once the batch is solved (no host round trip per block)
            self._iterate_fused(0, seeds)
            self._stop_flag.zero_()
            self._blocks_run.zero_()
            needed = int(math.ceil(c.batch_success_threshold * P))
            for it in range(outer):
                self._iterate_fused(c.inner_iterations)
                if it < outer - 1:
                    if self._sharded():
                        self._global_batch_status(needed)
                    else:
                        linalg_hip.seed_ik_batch_status(self.success, P, S, needed, self._stop_flag)
        else:
            self._evaluate_candidate(seeds, initial=True)
        it = 0
        for it in range(outer if not fused else 0):
            self._run_inner()
            if it < outer - 1:  # reference _calculate_exit_condition (:452-468)
                solved = (self.success.view(P, S).sum(-1) >= 1)
                if self._sharded():  # a problem counts as solved when ANY rank holds a converged seed of it
                    from ..distributed import all_reduce_max

                    solved = all_reduce_max(solved.to(torch.int32))
                if int(solved.sum()) >= c.batch_success_threshold * P:
                    break
        self._last_outer = it
        if self._sharded():
            return self._rank_over_all_shards(return_seeds, current_position)
        if S <= 1024 and return_seeds <= S:  # one launch instead of ~20 torch kernels (masks, top-k, gathers)
            dev = self.device
            # the four outputs are views of ONE buffer, so that a caller that wants copies makes one (_unpack_like)
            self._select_pack, (ok_o, sol_o, pos_o, ori_o) = _packed_outputs(
                dev, [((P, return_seeds), torch.uint8), ((P, return_seeds, D), torch.float32), ((P, return_seeds), torch.float32),
                      ((P, return_seeds), torch.float32)])
            cur = None
            if c.start_cspace_dist_weight > 0 and current_position is not None:
                cur = current_position.to(dev, torch.float32).reshape(P, D).contiguous()
            linalg_hip.seed_ik_select(ok_o, sol_o, pos_o, ori_o, self.q.view(P, S, D), self.position_error, self.orientation_error,
                                      self._limits[0], self._limits[1], cur, c.position_tolerance, c.orientation_tolerance,
                                      c.start_cspace_dist_weight, c.joint_limit_weight > 0, return_seeds)
            return ok_o.view(torch.bool), sol_o, pos_o, ori_o  # (the kernel writes 0 / 1 bytes)
        pos, ori = self.position_error.view(P, S), self.orientation_error.view(P, S)
        q = self.q.view(P, S, D)
        ok = (pos < c.position_tolerance) & (ori < c.orientation_tolerance)
        if c.joint_limit_weight > 0:
            ok &= ((q > self._limits[0]) & (q < self._limits[1])).all(-1)
        # reference _select_top_solutions (:522-572)
        costs = pos + ori
        if c.start_cspace_dist_weight > 0 and current_position is not None:
            costs = costs + c.start_cspace_dist_weight * torch.norm(q - current_position.view(P, 1, D), dim=-1)
        costs = costs + 1e10 * (~ok).float()
        top = torch.topk(costs, k=return_seeds, dim=-1, largest=False).indices
        g = lambda t: torch.gather(t, 1, top)  # noqa: E731
        sol = torch.gather(q, 1, top.unsqueeze(-1).expand(P, return_seeds, D))
        return g(ok), sol, g(pos), g(ori)

    # ------------------------------------------------------------------ seed shards (one process per GPU)
    def _global_batch_status(self, needed: int) -> None:
        """the exit test of the fused path over ALL seed shards: one all-reduce (MAX) of the per-problem "has a converged
        seed" byte vector, enqueued on the stream like the launches around it (no host round trip); the flag is sticky,
        as in ``curobo_hip_seed_ik_batch_status``"""
        from ..distributed import all_reduce_max

        solved = all_reduce_max(self.success.view(self.P, self.S).amax(dim=1).to(torch.int32))
        self._stop_flag.copy_(torch.maximum(self._stop_flag, (solved.sum() >= needed).to(torch.int32).view(1)))

    def _rank_over_all_shards(self, return_seeds: int, current_position: Optional[torch.Tensor]):
        """reference _select_top_solutions (:522-572) over the LM runs of every rank: the ``return_seeds`` best of the
        global seed set, ordered by (cost, global seed index) -- identical on every rank and for every world size"""
        from ..distributed import global_topk

        P, S, D, c = self.P, self.S, self.D, self.cfg
        pos, ori = self.position_error.view(P, S), self.orientation_error.view(P, S)
        q = self.q.view(P, S, D)
        ok = (pos < c.position_tolerance) & (ori < c.orientation_tolerance)
        if c.joint_limit_weight > 0:
            ok &= ((q > self._limits[0]) & (q < self._limits[1])).all(-1)
        costs = pos + ori
        if c.start_cspace_dist_weight > 0 and current_position is not None:
            costs = costs + c.start_cspace_dist_weight * torch.norm(q - current_position.to(self.device).view(P, 1, D), dim=-1)
        costs = costs + 1e10 * (~ok).float()
        payload = torch.cat([q, pos.unsqueeze(-1), ori.unsqueeze(-1), ok.float().unsqueeze(-1)], dim=-1)
        _, _, best = global_topk(costs, payload, self.seed_offset, return_seeds)
        return best[..., D + 2] > 0.5, best[..., :D].contiguous(), best[..., D], best[..., D + 1]
