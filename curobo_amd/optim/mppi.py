"""MPPI: sampling-based optimiser with softmax-weighted distribution updates.

Mirrors the reference's ``MPPI`` / ``ParticleOptCore`` iteration
(``curobo/_src/optim/particle/mppi.py:173-313``, ``optim/components/particle_opt_core.py:283-470``):

    for each iteration:
        actions = squash(mean + noise * scale_tril)  (+ negated-mean and null particles)
        costs   = rollout(actions)                   (num_problems * particles rollouts, cost only)
        mean, cov, scale_tril, best <- softmax-weighted moments of the particles

The distribution update is one HIP launch (``backends.optimization.mppi_update_distribution``)
instead of the reference's chain of torch kernels.  The particle noise is the reference's sample library
(``optim/particle_samples.py``: scrambled Halton points through the 2000-row buffer, erfinv, three-tap filter; optional
STOMP share), generated once per optimiser for the GLOBAL problem set and cycled as
``GaussianDistribution.get_samples`` cycles it: the same seed gives the reference's particle set
(``tests/test_mppi_samples.py``).  ``noise="torch"`` draws fresh Gaussian noise from a ``torch.Generator`` instead.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch

from ..backends import optimization as optimization_hip


@dataclass
class MPPICfg:
    """Names and defaults follow ``MPPICfg`` (``optim/particle/mppi.py:64-140``)."""

    num_problems: int = 1
    num_particles: int = 256
    num_iters: int = 100
    inner_iters: int = 1
    gamma: float = 1.0
    beta: float = 0.1
    init_cov: float = 0.5
    step_size_mean: float = 0.9
    step_size_cov: float = 0.1
    kappa: float = 0.01
    null_act_frac: float = 0.0
    update_cov: bool = True
    sample_mode: str = "MEAN"  # MEAN | BEST
    seed: int = 0
    # sample_params (reference ParticleSamplerCfg, sample_strategies/particle_sampler_cfg.py:18-44)
    noise: str = "sample_lib"  # "sample_lib" = the reference's library (below) | "torch" = fresh Gaussian noise every iteration
    sample_ratio: Optional[dict] = None  # default {"halton": 1.0}; e.g. {"halton": 0.5, "stomp": 0.5}
    filter_coeffs: Optional[Tuple[float, float, float]] = (0.3, 0.3, 0.4)
    fixed_samples: bool = True      # one noise set for every iteration (else num_iters sets, cycled)
    sample_per_problem: bool = True  # every problem its own particles (else one set repeated over the problems)

    @staticmethod
    def reference_task(task: str, **overrides) -> "MPPICfg":
        """the optimiser block of the reference's particle stage files: ``task`` = "ik" (content/configs/task/ik/particle_ik.yml: 4
        iterations of 25 Halton particles, BEST) or "trajopt" (task/trajopt/particle_trajopt.yml: 3 iterations of 25 STOMP particles,
        BEST) -- the short particle stage the reference can put in front of its L-BFGS stage (``MultiStageOptimizer``).  Held to the
        files by ``tests/test_types_members.py::test_mppi_reference_tasks_are_the_reference_files``."""
        if task == "ik":
            base = dict(gamma=1.0, init_cov=1.0, kappa=0.01, beta=1.0, step_size_cov=0.2, step_size_mean=0.9, num_iters=4, inner_iters=4,
                        null_act_frac=0.0, num_particles=25, sample_mode="BEST", filter_coeffs=(0.0, 0.0, 1.0), fixed_samples=True,
                        sample_ratio={"halton": 1.0}, sample_per_problem=True, seed=0, update_cov=True)
        elif task == "trajopt":
            base = dict(gamma=1.0, init_cov=0.5, kappa=0.001, beta=0.01, step_size_cov=0.1, step_size_mean=0.98, num_iters=3, inner_iters=3,
                        null_act_frac=0.0, num_particles=25, sample_mode="BEST", filter_coeffs=(0.3, 0.3, 0.4), fixed_samples=True,
                        sample_ratio={"stomp": 1.0}, sample_per_problem=True, seed=0, update_cov=True)
        else:
            raise ValueError(f"task must be 'ik' or 'trajopt', got {task!r}")
        return MPPICfg(**{**base, **overrides})


class MPPI:
    """``optimize(seed[num_problems, action_horizon, action_dim])`` -> action of the fitted distribution.

    ``cost_fn(actions[num_problems * num_particles, action_horizon * action_dim])`` returns either
    total costs ``[N]`` or per-step costs ``[N, horizon]`` (discounted with ``gamma``).
    """

    def __init__(self, cfg: MPPICfg, cost_fn: Callable[[torch.Tensor], torch.Tensor], action_horizon: int,
                 action_dim: int, action_bounds: Tuple[torch.Tensor, torch.Tensor], device, problem_offset: int = 0,
                 global_num_problems: Optional[int] = None):
        """PROBLEM shard (one process per GPU; SURVEY.md section 8e: the softmax couples the particles of a problem, so the
        particle stage shards by problem): this instance optimises problems ``[problem_offset, problem_offset +
        cfg.num_problems)`` of ``global_num_problems``.  The particle noise of a problem depends on its GLOBAL index only
        (every rank draws the global noise tensor from the same generator state and keeps its rows), so any world size
        samples the same particles; results are assembled with ``distributed.all_gather_problems``.  No collective inside
        the iteration."""
        self.cfg, self.cost_fn = cfg, cost_fn
        self.problem_offset = int(problem_offset)
        self.global_num_problems = int(global_num_problems) if global_num_problems is not None else cfg.num_problems
        if not (0 <= self.problem_offset and self.problem_offset + cfg.num_problems <= self.global_num_problems):
            raise ValueError(f"problem shard [{self.problem_offset}, {self.problem_offset + cfg.num_problems}) is not inside "
                             f"the {self.global_num_problems} problems of the job")
        self.action_horizon, self.action_dim, self.device = action_horizon, action_dim, device
        B, P, Ha, D = cfg.num_problems, cfg.num_particles, action_horizon, action_dim
        lows, highs = action_bounds
        self._low = lows.to(device=device, dtype=torch.float32).view(1, 1, 1, D)
        self._high = highs.to(device=device, dtype=torch.float32).view(1, 1, 1, D)
        # particle layout of the reference (particle_opt_core.py:190-216): sampled, then one
        # negated-mean particle and the null (zero-action) particles
        self.null_per_problem = int(round(cfg.null_act_frac * P))
        self.neg_per_problem = 1 if self.null_per_problem > 0 else 0
        self.sampled_per_problem = P - self.null_per_problem - self.neg_per_problem
        z = lambda *s: torch.zeros(*s, device=device, dtype=torch.float32)  # noqa: E731
        self.mean, self.cov, self.scale_tril = z(B, Ha, D), z(B, 1, D), z(B, 1, D)
        self._new_mean, self._new_cov, self._new_tril = z(B, Ha, D), z(B, 1, D), z(B, 1, D)
        self.best_traj, self.weights = z(B, Ha, D), z(B, P)
        self.actions = z(B, P, Ha, D)
        self._gen = torch.Generator(device=device)
        self._gamma_seq: Optional[torch.Tensor] = None
        self._sample_set: Optional[torch.Tensor] = None
        self._sample_iter = 0
        if cfg.noise not in ("sample_lib", "torch"):
            raise ValueError(f"MPPICfg.noise must be 'sample_lib' or 'torch', got {cfg.noise!r}")
        if cfg.noise == "sample_lib" and self.sampled_per_problem > 0:
            from .particle_samples import ParticleSampleLib, sample_set

            # the job's set (global problem count), this shard's problems of it: any world size samples the same particles
            lib = ParticleSampleLib(Ha, D, seed=cfg.seed, sample_ratio=cfg.sample_ratio, filter_coeffs=cfg.filter_coeffs)
            full = sample_set(lib, self.global_num_problems, self.sampled_per_problem, 1 if cfg.fixed_samples else cfg.num_iters,
                              cfg.sample_per_problem)
            self._sample_set = full[:, self.problem_offset:self.problem_offset + B].to(device).contiguous()
        self.reset_distribution()

    def reset_distribution(self) -> None:
        self.mean.zero_()
        self.cov.fill_(self.cfg.init_cov)
        self.scale_tril.copy_(torch.sqrt(self.cov))
        self._gen.manual_seed(self.cfg.seed)
        self._sample_iter = 0

    # reference ParticleOptCore.sample_actions (:393-442), CLAMP squash
    @torch.no_grad()
    def sample_actions(self) -> torch.Tensor:
        B, Ha, D = self.cfg.num_problems, self.action_horizon, self.action_dim
        n = self.sampled_per_problem
        if self._sample_set is not None:  # reference GaussianDistribution.get_samples (:243-256)
            noise = self._sample_set[self._sample_iter]
            self._sample_iter = (self._sample_iter + 1) % self._sample_set.shape[0]
        elif self.global_num_problems == B:
            noise = torch.randn(B, n, Ha, D, device=self.device, generator=self._gen)
        else:  # a problem shard: the job's noise tensor, this rank's problems of it
            noise = torch.randn(self.global_num_problems, n, Ha, D, device=self.device, generator=self._gen)
            noise = noise[self.problem_offset:self.problem_offset + B]
        self.actions[:, :n] = self.mean.unsqueeze(1) + noise * self.scale_tril.unsqueeze(1)
        if self.neg_per_problem:
            self.actions[:, n:n + 1] = -self.mean.unsqueeze(1)
        if self.null_per_problem:
            self.actions[:, n + self.neg_per_problem:] = 0.0
        torch.maximum(self.actions, self._low, out=self.actions)
        torch.minimum(self.actions, self._high, out=self.actions)
        return self.actions

    @torch.no_grad()
    def update_distribution(self, costs: torch.Tensor) -> None:
        """reference MPPI._update_distribution (:201-262); ``costs`` [B*P] or [B*P, horizon]"""
        cfg, B, P = self.cfg, self.cfg.num_problems, self.cfg.num_particles
        costs = costs.reshape(B, P, -1).contiguous()
        hc = costs.shape[-1]
        if self._gamma_seq is None or self._gamma_seq.shape[0] != hc:
            self._gamma_seq = torch.cumprod(torch.full((hc,), cfg.gamma, device=self.device), 0) / cfg.gamma
        optimization_hip.mppi_update_distribution(
            self._new_mean, self._new_cov, self._new_tril, self.best_traj, self.weights, costs, self._gamma_seq,
            self.actions, self.mean, self.cov, cfg.beta, cfg.step_size_mean, cfg.step_size_cov if cfg.update_cov else 0.0,
            cfg.kappa if cfg.update_cov else 0.0)
        self.mean, self._new_mean = self._new_mean, self.mean
        if cfg.update_cov:
            self.cov, self._new_cov = self._new_cov, self.cov
            self.scale_tril, self._new_tril = self._new_tril, self.scale_tril

    def _opt_step(self) -> torch.Tensor:
        acts = self.sample_actions()
        costs = self.cost_fn(acts.view(-1, self.action_horizon * self.action_dim))
        self.update_distribution(costs)
        return costs

    @torch.no_grad()
    def optimize(self, seed_action: torch.Tensor) -> torch.Tensor:
        """reference ParticleOptCore.optimize (:283-312): seed the mean, iterate, return the action"""
        self.mean.copy_(seed_action.reshape(self.mean.shape))
        costs = None
        for _ in range(self.cfg.num_iters):
            costs = self._opt_step()
        self.last_costs = costs
        return (self.best_traj if self.cfg.sample_mode == "BEST" else self.mean).clone()
