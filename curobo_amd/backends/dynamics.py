"""HIP backend for the RNEA inverse-dynamics kernels.

Drop-in for ``curobo/_src/curobolib/backends/cuda_core_backend/dynamics.py`` (same function names
and positional argument order: ``launch_rnea_forward`` :24-131, ``launch_rnea_backward`` :134-260).
``forward_cache`` is the reference's opaque per-element scratch ``[batch, num_links*20]``; only its
size is part of the contract, the HIP kernels use a ``[link][20][batch]`` layout inside it.
"""

from __future__ import annotations

import weakref

from typing import Dict, Optional, Tuple

import torch

from .._lib import check, current_stream, load, ptr

_workspaces: Dict[Tuple[str, int], torch.Tensor] = {}
_dfs_orders: Dict[Tuple[int, int, str], Tuple["weakref.ref", torch.Tensor]] = {}


def _walk_order(link_map: torch.Tensor, level_links: torch.Tensor) -> torch.Tensor:
    """Depth-first (parents first) order of the links as int16, cached per link table.  The kernels walk the tree
    serially per element; any parents-first order is correct, and in depth-first order the parent of a link is mostly
    the link just processed, whose state the kernels then keep in registers instead of re-reading it from the cache.
    One host read-back per robot: call once outside graph capture (the first, warm-up call does).  The entry lives as
    long as ``link_map`` does: pass the robot's own table (``KinematicsParams.link_map``), not a temporary copy, and keep
    the returned tensor referenced while a struct holds its raw pointer (the fused trajopt terms)."""
    key = (link_map.data_ptr(), int(link_map.numel()), str(link_map.device))
    hit = _dfs_orders.get(key)
    # an entry only counts while the tensor it was made from is alive: the caching allocator hands a freed address
    # to the next table of the same size (another robot), and a stale order need not be parents-first for it
    order = hit[1] if hit is not None and hit[0]() is not None else None
    if order is None:
        par = link_map.detach().cpu().numpy().astype(int)
        n = par.shape[0]
        kids = [[] for _ in range(n)]
        roots = []
        for k in range(n):
            if par[k] < 0 or par[k] == k:
                roots.append(k)
            else:
                kids[par[k]].append(k)
        out, stack = [], roots[::-1]
        while stack:
            k = stack.pop()
            out.append(k)
            stack.extend(kids[k][::-1])
        if len(out) != n:  # not a forest rooted as expected: keep the caller's level order
            order = level_links
        else:
            order = torch.as_tensor(out, dtype=level_links.dtype).to(level_links.device)
        _dfs_orders[key] = (weakref.ref(link_map), order)
        from .geometry import _evict_dead

        weakref.finalize(link_map, _evict_dead, _dfs_orders, key)  # no dead entries (and their device tensors) pile up
    return order


def _workspace(device: torch.device, n_floats: int) -> torch.Tensor:
    """Adjoint scratch of the backward kernel (f_bar, a_bar, v_bar per link and element).  The
    reference keeps these in shared memory; here they are a per-device tensor that only ever grows.
    The first backward call at a new size allocates, so warm up once before hipGraph capture."""
    key = (str(device), 0)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < n_floats:
        ws = torch.empty(n_floats, device=device, dtype=torch.float32)
        _workspaces[key] = ws
    return ws


def launch_rnea_forward(
    tau: torch.Tensor,
    q: torch.Tensor,
    qd: torch.Tensor,
    qdd: torch.Tensor,
    fixed_transforms: torch.Tensor,
    link_masses_com: torch.Tensor,
    link_inertias: torch.Tensor,
    joint_map_type: torch.Tensor,
    joint_map: torch.Tensor,
    link_map: torch.Tensor,
    joint_offset_map: torch.Tensor,
    gravity: torch.Tensor,
    level_starts: torch.Tensor,
    level_links: torch.Tensor,
    forward_cache: torch.Tensor,
    batch_size: int,
    num_links: int,
    num_dof: int,
    n_levels: int,
    threads_per_batch: int = 1,
    f_ext: Optional[torch.Tensor] = None,
    scratch: Optional[torch.Tensor] = None,
):
    """``scratch`` (extension, 3 * num_dof * batch_size floats): the launch transposes its inputs into it and the walk reads them
    from there instead of staging them through LDS (``curobo_hip_launch_rnea_forward_scratch``: same values)"""
    if forward_cache.numel() < batch_size * num_links * 20:
        raise ValueError("forward_cache must hold batch_size * num_links * 20 floats")
    if scratch is not None:
        if scratch.numel() < 3 * num_dof * batch_size:
            raise ValueError("scratch must hold 3 * num_dof * batch_size floats")
        check(load().curobo_hip_launch_rnea_forward_scratch(
            ptr(tau), ptr(q), ptr(qd), ptr(qdd), ptr(fixed_transforms), ptr(link_masses_com), ptr(link_inertias),
            ptr(joint_map_type), ptr(joint_map), ptr(link_map), ptr(joint_offset_map), ptr(gravity), ptr(level_starts),
            ptr(_walk_order(link_map, level_links)), ptr(forward_cache), batch_size, num_links, num_dof, n_levels, threads_per_batch,
            ptr(f_ext), ptr(scratch), current_stream(tau),
        ))
        return
    check(load().curobo_hip_launch_rnea_forward(
        ptr(tau), ptr(q), ptr(qd), ptr(qdd), ptr(fixed_transforms), ptr(link_masses_com), ptr(link_inertias),
        ptr(joint_map_type), ptr(joint_map), ptr(link_map), ptr(joint_offset_map), ptr(gravity), ptr(level_starts),
        ptr(_walk_order(link_map, level_links)), ptr(forward_cache), batch_size, num_links, num_dof, n_levels, threads_per_batch,
        ptr(f_ext), current_stream(tau),
    ))


def launch_rnea_backward(
    grad_q: torch.Tensor,
    grad_qd: torch.Tensor,
    grad_qdd: torch.Tensor,
    grad_tau: torch.Tensor,
    q: torch.Tensor,
    qd: torch.Tensor,
    fixed_transforms: torch.Tensor,
    link_masses_com: torch.Tensor,
    link_inertias: torch.Tensor,
    joint_map_type: torch.Tensor,
    joint_map: torch.Tensor,
    link_map: torch.Tensor,
    joint_offset_map: torch.Tensor,
    gravity: torch.Tensor,
    level_starts: torch.Tensor,
    level_links: torch.Tensor,
    forward_cache: torch.Tensor,
    batch_size: int,
    num_links: int,
    num_dof: int,
    n_levels: int,
    threads_per_batch: int = 1,
    grad_f_ext: Optional[torch.Tensor] = None,
    workspace: Optional[torch.Tensor] = None,
    scratch: Optional[torch.Tensor] = None,
    scratch_holds_q_qd: bool = False,
    accumulate: bool = False,
):
    """Gradient buffers are fully rewritten (no ``zero_()`` needed), as in the reference.  ``scratch``: see ``launch_rnea_forward``;
    ``scratch_holds_q_qd``: it still holds q and qd of the forward launch on the same inputs (only grad_tau is transposed);
    ``accumulate`` (scratch launches only): add to the gradient buffers instead of rewriting them."""
    need = batch_size * num_links * 18
    ws = workspace if workspace is not None else _workspace(grad_q.device, need)
    if ws.numel() < need:
        raise ValueError("workspace must hold batch_size * num_links * 18 floats")
    if accumulate and scratch is None:
        raise ValueError("accumulate needs the scratch form of the launch")
    if scratch is not None:
        if scratch.numel() < 3 * num_dof * batch_size:
            raise ValueError("scratch must hold 3 * num_dof * batch_size floats")
        check(load().curobo_hip_launch_rnea_backward_scratch(
            ptr(grad_q), ptr(grad_qd), ptr(grad_qdd), ptr(grad_tau), ptr(q), ptr(qd), ptr(fixed_transforms),
            ptr(link_masses_com), ptr(link_inertias), ptr(joint_map_type), ptr(joint_map), ptr(link_map),
            ptr(joint_offset_map), ptr(gravity), ptr(level_starts), ptr(_walk_order(link_map, level_links)), ptr(forward_cache), batch_size,
            num_links, num_dof, n_levels, threads_per_batch, ptr(grad_f_ext), ptr(ws), ptr(scratch), int(bool(scratch_holds_q_qd)) | (2 if accumulate else 0),
            current_stream(grad_q),
        ))
        return
    check(load().curobo_hip_launch_rnea_backward(
        ptr(grad_q), ptr(grad_qd), ptr(grad_qdd), ptr(grad_tau), ptr(q), ptr(qd), ptr(fixed_transforms),
        ptr(link_masses_com), ptr(link_inertias), ptr(joint_map_type), ptr(joint_map), ptr(link_map),
        ptr(joint_offset_map), ptr(gravity), ptr(level_starts), ptr(_walk_order(link_map, level_links)), ptr(forward_cache), batch_size,
        num_links, num_dof, n_levels, threads_per_batch, ptr(grad_f_ext), ptr(ws), current_stream(grad_q),
    ))
