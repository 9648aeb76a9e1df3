"""Knots of a trajectory that brings a moving robot to rest: the reference's deceleration seeds (``TrajectorySeedGenerator.
generate_deceleration_seeds``, util/trajectory_seed_generator.py:122-376), restated.  An acceleration profile that opposes the current
velocity -- its magnitude decays from 10 rad/s^2 over the horizon (linear / exponential / cosine), blended in from the CURRENT
acceleration over the first few knots -- is integrated twice with explicit Euler steps; a joint whose velocity would change sign has
stopped and stays where it is.  Pure torch, no device work of its own; used by ``MPCSolver.prepare_safe_deceleration_trajectory`` (a
utility in the reference too: its control loop has the fallback switched off, solver_mpc.py:679)."""

from __future__ import annotations

import math

import torch

MAX_DECELERATION = 10.0  # rad/s^2 or m/s^2 (the reference's constant)


def deceleration_magnitudes(n: int, profile: str, device=None, dtype=torch.float32) -> torch.Tensor:
    """[n] magnitude of the opposing acceleration at the knots: MAX at the first, decaying to (nearly) nothing at the last"""
    t = torch.linspace(0, 1, n, device=device, dtype=dtype)
    if profile == "linear":
        return MAX_DECELERATION * (1.0 - t)
    if profile == "smooth":
        return MAX_DECELERATION * (torch.cos(t * math.pi) + 1.0) / 2.0
    return MAX_DECELERATION * torch.exp(-3.0 * t)  # "exponential", and the fallback for an unknown name as in the reference


def deceleration_accelerations(velocity: torch.Tensor, acceleration: torch.Tensor, n: int, profile: str = "exponential") -> torch.Tensor:
    """[batch, n, dof] accelerations: (1 - b_t) a_now + b_t target_t over the first min(5, n // 3) knots (b from 0 to 1), clamped to
    the maximum, the target -sign(v) magnitude_t after them; zero for joints that do not move"""
    moving = velocity.abs() > 1e-6
    target = (-torch.sign(velocity)).unsqueeze(1) * deceleration_magnitudes(n, profile, velocity.device, velocity.dtype).view(1, n, 1)
    steps = min(5, n // 3)
    out = target.clone()
    if steps > 0:
        b = (torch.arange(steps, device=velocity.device, dtype=velocity.dtype) / max(steps - 1, 1)).view(1, steps, 1)
        blended = (1.0 - b) * acceleration.unsqueeze(1) + b * target[:, :steps]
        out[:, :steps] = torch.sign(blended) * blended.abs().clamp(0, MAX_DECELERATION)
    return torch.where(moving.unsqueeze(1), out, torch.zeros_like(out))


def deceleration_knots(position: torch.Tensor, velocity: torch.Tensor, acceleration: torch.Tensor, dt: float, n: int,
                       profile: str = "exponential") -> torch.Tensor:
    """[batch, n, dof]: p_0 = the current position, p_{t+1} = p_t + v_t dt, v_{t+1} = v_t + a_t dt -- set to zero for good once it is
    tiny or points against the initial velocity"""
    acc = deceleration_accelerations(velocity, acceleration, n, profile)
    moving, sign0 = velocity.abs() > 1e-6, torch.sign(velocity)
    p, v, out = position, velocity, [position]
    for t in range(1, n):
        nv = v + acc[:, t - 1] * dt
        nv = torch.where(((torch.sign(nv) != sign0) & moving) | (nv.abs() < 1e-6), torch.zeros_like(nv), nv)
        p = p + v * dt
        v = nv
        out.append(p)
    return torch.stack(out, dim=1)
