"""Scene-collision autograd functions -- reference ``geom/collision/wp_autograd.py:37-249``
(NVIDIA Warp there, one HIP launch here) and ``CollisionBuffer``
(``geom/collision/buffer_collision.py:25-105``)."""

from __future__ import annotations

from dataclasses import dataclass

import torch

from ..backends import collision as collision_hip
from .tensor_checks import check_float32_tensors, check_int32_tensors


@dataclass
class CollisionBuffer:
    """distance [B,H,S] and gradient [B,H,S,4] written by the collision kernel."""

    distance: torch.Tensor
    gradient: torch.Tensor

    @staticmethod
    def create(batch: int, horizon: int, num_spheres: int, device) -> "CollisionBuffer":
        return CollisionBuffer(torch.zeros(batch, horizon, num_spheres, device=device),
                               torch.zeros(batch, horizon, num_spheres, 4, device=device))

    def zero_(self) -> None:  # kept for API parity; the HIP kernel rewrites both buffers fully
        self.distance.zero_()
        self.gradient.zero_()


def _launch(query_spheres, buffer, scene, weight, activation_distance, env_query_idx, use_multi_env, sweep,
            speed_dt=None, enable_speed_metric=False):
    b, h, n, _ = query_spheres.shape
    device = query_spheres.device
    check_float32_tensors(device, query_spheres=query_spheres, distance=buffer.distance, gradient=buffer.gradient,
                          weight=weight, activation_distance=activation_distance, speed_dt=speed_dt)
    check_int32_tensors(device, env_query_idx=env_query_idx)
    collision_hip.sphere_obstacle_collision(
        buffer.distance, buffer.gradient, query_spheres.detach(), scene.struct, weight, activation_distance,
        env_query_idx, b, h, n, bool(use_multi_env), 3 if sweep else 0, bool(enable_speed_metric), speed_dt)


def _backward(ctx, grad_output):
    grad_sph = None
    if ctx.needs_input_grad[0]:
        (grad_buffer,) = ctx.saved_tensors
        grad_sph = grad_buffer
        if ctx.return_loss:
            grad_sph = grad_buffer * grad_output.unsqueeze(-1)
    return grad_sph


class SphereObstacleCollision(torch.autograd.Function):
    """reference wp_autograd.py:37-121 (``max_distance`` is accepted for signature parity; the
    voxel max distance travels inside the scene struct)."""

    @staticmethod
    def forward(ctx, query_spheres, buffer: CollisionBuffer, scene, weight, activation_distance, max_distance,
                env_query_idx, use_multi_env: bool, return_loss: bool = False):
        _launch(query_spheres, buffer, scene, weight, activation_distance, env_query_idx, use_multi_env, False)
        ctx.return_loss = return_loss
        ctx.save_for_backward(buffer.gradient)
        return buffer.distance

    @staticmethod
    def backward(ctx, grad_output):
        return (_backward(ctx, grad_output),) + (None,) * 8


class SweptSphereObstacleCollision(torch.autograd.Function):
    """reference wp_autograd.py:124-249"""

    @staticmethod
    def forward(ctx, query_spheres, buffer: CollisionBuffer, scene, weight, activation_distance, max_distance,
                speed_dt, enable_speed_metric: bool, env_query_idx, use_multi_env: bool, return_loss: bool = False):
        _launch(query_spheres, buffer, scene, weight, activation_distance, env_query_idx, use_multi_env, True,
                speed_dt, enable_speed_metric)
        ctx.return_loss = return_loss
        ctx.save_for_backward(buffer.gradient)
        return buffer.distance

    @staticmethod
    def backward(ctx, grad_output):
        return (_backward(ctx, grad_output),) + (None,) * 10
