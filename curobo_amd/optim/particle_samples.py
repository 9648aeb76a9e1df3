"""The noise of the particle optimisers: the reference's sample library, restated.

Reference: ``curobo/_src/optim/particle/sample_strategies/particle_sampler.py`` (``MixedParticleSampler`` /
``ParticleSampler``), ``util/sampling/sample_buffer.py:60-215`` (``SampleBuffer``: 2000 scrambled-Halton points, rows drawn
from them by a seeded integer generator, the erfinv map to a Gaussian), ``processor_standard.py:69-87`` (the recursive
three-tap filter), ``processor_stomp.py`` + ``stomp_covariance.py`` (STOMP-correlated noise), and
``optim/components/gaussian_distribution.py:170-258`` (the pre-generated sample set an optimiser cycles through).

Everything here runs ONCE per optimiser (``fixed_samples``: one set for all iterations, as in the reference's default
``ParticleSamplerCfg``), on the host; the iteration itself only reads the device copy.  The integer generator that picks
the rows of the Halton buffer is the CPU one: the reference uses the generator of whatever device it runs on, whose CUDA
and CPU streams already differ from each other -- the CPU stream is the one a golden produced in the build container can
pin (``tests/golden/mppi_samples_golden.npz``), and it makes the particles independent of the device and of the world
size (a problem shard slices the job's set by GLOBAL problem index).
"""

from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch


def stomp_scale_tril(horizon: int, stencil_type: str = "3point") -> torch.Tensor:
    """Cholesky factor of the normalised STOMP covariance (A^T A)^-1 of a finite-difference stencil, boundaries zeroed
    (reference ``get_stomp_cov``, stomp_covariance.py:17-120); float32 on the CPU like the reference."""
    stencils = {"3point": [0.0, 0.0, 1.0, -2.0, 1.0, 0.0, 0.0],
                "5point": [0.0, -1 / 12, 4 / 3, -5 / 2, 4 / 3, -1 / 12, 0.0],
                "7point": [1 / 90, -3 / 20, 3 / 2, -49 / 18, 3 / 2, -3 / 20, 1 / 90]}
    if stencil_type not in stencils:
        raise ValueError(f"Unknown stencil_type: {stencil_type}")
    A = torch.zeros((horizon, horizon), dtype=torch.float32)
    for k, coeff in enumerate(torch.tensor(stencils[stencil_type], dtype=torch.float32)):
        if coeff != 0:
            off = k - 3
            rows = torch.arange(horizon - off) if off >= 0 else torch.arange(-off, horizon)
            cols = rows + off
            A[rows, torch.clamp(cols, 0, horizon - 1)] = coeff
    M = torch.inverse(torch.matmul(A.T, A))
    M[0, :] = 0.0
    M[:, 0] = 0.0
    M[horizon - 1, :] = 0.0
    M[:, horizon - 1] = 0.0
    M[0, 0] = 1e-8
    M[horizon - 1, horizon - 1] = 1e-8
    cov = M / (torch.max(torch.abs(M)) + 1e-8)
    cov = (cov + cov.T) / 2
    try:
        if (cov == cov.T).all() and (torch.linalg.eigvals(cov).real >= 0).all():
            return torch.linalg.cholesky(cov)
    except RuntimeError:
        pass
    return cov


class _HaltonGaussian:
    """``SampleBuffer(HaltonSequencer(ndims, seed), store_buffer=2000).get_gaussian_samples``"""

    def __init__(self, ndims: int, seed: int, store_buffer: int = 2000):
        from scipy.stats.qmc import Halton

        self.ndims = ndims
        self.buffer = torch.tensor(Halton(d=ndims, seed=seed, scramble=True).random(store_buffer), dtype=torch.float32)
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        self._state0 = self.gen.get_state().clone()

    def reset(self) -> None:
        self.gen.set_state(self._state0)

    def gaussian(self, n: int) -> torch.Tensor:
        index = torch.randint(0, self.buffer.shape[0], (n,), generator=self.gen)
        u = self.buffer[index]
        # (1.98 u - 0.99 keeps erfinv finite at the ends of the unit interval: sample_buffer.py:209-216)
        g = torch.sqrt(torch.tensor([2.0])) * torch.erfinv(1.98 * u - 0.99)
        return torch.matmul(g, torch.eye(self.ndims))


class _RandomGaussian:
    """``SampleBuffer(RandomSequencer(ndims, seed), store_buffer=None)``: numpy's default_rng uniform stream"""

    def __init__(self, ndims: int, seed: int):
        self.ndims, self.seed = ndims, seed
        self.rng = np.random.default_rng(seed)

    def reset(self) -> None:
        self.rng = np.random.default_rng(self.seed)

    def gaussian(self, n: int) -> torch.Tensor:
        u = torch.tensor(self.rng.random((n, self.ndims)), dtype=torch.float32)
        return torch.sqrt(torch.tensor([2.0])) * torch.erfinv(1.98 * u - 0.99)


class ParticleSampleLib:
    """``get_samples(n)`` -> noise [n, horizon, action_dim] (CPU float32), mixed by ``sample_ratio`` in the dict's order
    (reference ``MixedParticleSampler.get_samples``: per type ``round(n * ratio)`` samples, concatenated)."""

    def __init__(self, horizon: int, action_dim: int, seed: int = 0, sample_ratio: Optional[Dict[str, float]] = None,
                 filter_coeffs: Optional[Sequence[float]] = (0.3, 0.3, 0.4), stencil_type: str = "3point"):
        self.horizon, self.action_dim = horizon, action_dim
        self.sample_ratio = dict(sample_ratio) if sample_ratio is not None else {"halton": 1.0}
        self.filter_coeffs = None if filter_coeffs is None else tuple(float(v) for v in filter_coeffs)
        self._gens, self._tril = {}, None
        nd = horizon * action_dim
        for kind, ratio in self.sample_ratio.items():
            if ratio <= 0.0:
                continue
            if kind in ("halton", "stomp"):
                self._gens[kind] = _HaltonGaussian(nd, seed)
            elif kind == "random":
                self._gens[kind] = _RandomGaussian(nd, seed)
            else:
                raise ValueError(f"sample type {kind!r} is not supported (halton, stomp, random)")
        if "stomp" in self._gens:
            self._tril = stomp_scale_tril(horizon, stencil_type)

    def reset_seed(self) -> None:
        for g in self._gens.values():
            g.reset()

    def _filter(self, eps: torch.Tensor) -> torch.Tensor:
        if self.filter_coeffs is not None:  # processor_standard.py:69-87: recursive in time, in place
            b0, b1, b2 = self.filter_coeffs
            for i in range(2, eps.shape[1]):
                eps[:, i, :] = b0 * eps[:, i, :] + b1 * eps[:, i - 1, :] + b2 * eps[:, i - 2, :]
        return eps

    def _stomp(self, eps: torch.Tensor) -> torch.Tensor:
        if min(eps.shape) == 0:
            return eps
        out = torch.matmul(self._tril, eps)  # [H, H] x [n, H, D]
        m = torch.max(torch.abs(out))
        if m > 0:
            out = out / m
        out[:, 0, :] = 0.0
        out[:, -2:, :] = 0.0
        return out

    def get_samples(self, n: int) -> torch.Tensor:
        parts = []
        for kind, ratio in self.sample_ratio.items():
            k = round(n * ratio)
            if ratio == 0.0 or kind not in self._gens or k == 0:
                continue
            raw = self._gens[kind].gaussian(k).view(k, self.horizon, self.action_dim)
            parts.append(self._stomp(raw) if kind == "stomp" else self._filter(raw))
        if not parts:
            return torch.zeros(n, self.horizon, self.action_dim)
        return torch.cat(parts, dim=0)


def sample_set(lib: ParticleSampleLib, num_problems: int, particles: int, iters: int = 1, sample_per_problem: bool = True) -> torch.Tensor:
    """reference ``GaussianDistribution.initialize_samples`` (:170-204): [iters, num_problems, particles, H, D]; the last
    sampled particle of every problem carries no noise (it evaluates the mean itself)."""
    H, D = lib.horizon, lib.action_dim
    if sample_per_problem:
        s = lib.get_samples(particles * num_problems * iters).view(iters, num_problems, particles, H, D).clone()
    else:
        s = lib.get_samples(iters * particles).view(iters, 1, particles, H, D).repeat(1, num_problems, 1, 1, 1).clone()
    s[:, :, -1, :, :] = 0.0
    return s
