"""``torch.autograd.Function`` for inverse dynamics, mirroring ``RNEAForwardFunction``
(``curobo/_src/curobolib/cuda_ops/dynamics.py:94-345``): same positional inputs, the forward
writes ``tau`` and the per-element cache, the backward returns the three pre-allocated gradient
buffers (fully rewritten by the kernel) and, when requested, the external-force gradient."""

from __future__ import annotations

from typing import Optional

import torch

from ..backends import dynamics as dynamics_hip


class RNEAForwardFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, qd, qdd, tau, grad_q_buf, grad_qd_buf, grad_qdd_buf, forward_cache, fixed_transforms,
                link_masses_com, link_inertias, joint_map_type, joint_map, link_map, joint_offset_map, gravity,
                level_starts, level_links, num_links: int, num_dof: int, n_levels: int, threads_per_batch: int = 1,
                f_ext: Optional[torch.Tensor] = None, grad_f_ext_buf: Optional[torch.Tensor] = None):
        b = q.shape[0]
        dynamics_hip.launch_rnea_forward(
            tau, q, qd, qdd, fixed_transforms, link_masses_com, link_inertias, joint_map_type, joint_map, link_map,
            joint_offset_map, gravity, level_starts, level_links, forward_cache, b, num_links, num_dof, n_levels,
            threads_per_batch, f_ext)
        ctx.save_for_backward(q, qd, forward_cache)
        ctx.bufs = (grad_q_buf, grad_qd_buf, grad_qdd_buf, grad_f_ext_buf)
        ctx.consts = (fixed_transforms, link_masses_com, link_inertias, joint_map_type, joint_map, link_map,
                      joint_offset_map, gravity, level_starts, level_links)
        ctx.dims = (num_links, num_dof, n_levels, threads_per_batch)
        ctx.has_f_ext = f_ext is not None and grad_f_ext_buf is not None
        return tau

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_tau):
        q, qd, cache = ctx.saved_tensors
        gq, gqd, gqdd, gfe = ctx.bufs
        num_links, num_dof, n_levels, tpb = ctx.dims
        dynamics_hip.launch_rnea_backward(
            gq, gqd, gqdd, grad_tau.contiguous(), q, qd, *ctx.consts, cache, q.shape[0], num_links, num_dof, n_levels,
            tpb, gfe if ctx.has_f_ext else None)
        return (gq, gqd, gqdd) + (None,) * 19 + (gfe if ctx.has_f_ext else None, None)
