"""Seed-parallel multi-GPU layer: one process per GPU, seeds sharded, one tiny exchange.

The reference is single-GPU (no NCCL/MPI anywhere).  Every (problem, seed)
row is independent through transition, FK, costs, backward, L-BFGS and line search, so the seed
axis shards with NO data-path collective; the only exchange is the arg-min over seeds at the end
of a solve (reference single-GPU equivalent: ``solver/solver_ik.py:503-515`` top-k,
``solver/solver_trajopt.py:469-484``).  That exchange is ~KBs -> latency-bound, so it is a single
``all_gather`` (RCCL over xGMI; every peer is one hop away) of a packed
``[cost, global_seed_index, payload...]`` row per problem, after which every rank computes the
identical arg-min (ties -> lowest global seed index).
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(num_seeds: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous seed range [lo, hi) owned by ``rank`` (first ranks take the remainder)."""
    base, rem = divmod(num_seeds, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_gather_rows(flat: torch.Tensor, row: torch.Tensor, group: Optional[dist.ProcessGroup]) -> None:
    """``all_gather_into_tensor`` (RCCL over xGMI in production).  The gloo backend -- world-size-2 tests on CPU, or two
    ranks sharing one GPU -- has no device all-gather: device tensors are staged through the host there."""
    if row.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(flat.shape, dtype=flat.dtype)
        dist.all_gather_into_tensor(host, row.cpu().contiguous(), group=group)
        flat.copy_(host)
    else:
        dist.all_gather_into_tensor(flat, row.contiguous(), group=group)


def all_reduce_max(x: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """in-place MAX over the ranks (no-op alone); device tensors go through the host on the gloo backend"""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return x
    if x.is_cuda and dist.get_backend(group) == "gloo":
        host = x.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        x.copy_(host)
    else:
        dist.all_reduce(x, op=dist.ReduceOp.MAX, group=group)
    return x


def all_reduce_sum(x: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """in-place SUM over the ranks (no-op alone); device tensors go through the host on the gloo backend"""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return x
    if x.is_cuda and dist.get_backend(group) == "gloo":
        host = x.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        x.copy_(host)
    else:
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x


def broadcast_from_rank0(x: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """in-place broadcast of rank 0's ``x`` (no-op alone); device tensors go through the host on the gloo backend"""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return x
    if x.is_cuda and dist.get_backend(group) == "gloo":
        host = x.cpu()
        dist.broadcast(host, src=0, group=group)
        x.copy_(host)
    else:
        dist.broadcast(x, src=0, group=group)
    return x


def local_best(cost: torch.Tensor, payload: torch.Tensor, seed_offset: int) -> torch.Tensor:
    """cost[P, S_local], payload[P, S_local, V] -> packed [P, 2 + V] rows (cost, global idx, payload).

    The seed index travels as fp32 (exact below 2^24 seeds)."""
    if cost.is_cuda and cost.dtype == torch.float32 and payload.dtype == torch.float32:  # one launch instead of ~8
        from .backends import linalg as linalg_hip

        row = torch.empty(cost.shape[0], 2 + payload.shape[-1], device=cost.device, dtype=torch.float32)
        linalg_hip.argmin_rows(row, cost.contiguous(), payload.contiguous(), seed_offset)
        return row
    c, i = torch.min(cost, dim=1)  # torch.min returns the first minimal index on ties
    P = cost.shape[0]
    row = torch.empty(P, 2 + payload.shape[-1], device=cost.device, dtype=torch.float32)
    row[:, 0] = c
    row[:, 1] = (i + seed_offset).to(torch.float32)
    row[:, 2:] = payload[torch.arange(P, device=cost.device), i]
    return row


def global_argmin(cost: torch.Tensor, payload: torch.Tensor, seed_offset: int,
                  group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Best (cost[P], global_seed_idx[P], payload[P, V]) over the seeds of ALL ranks.

    Works unchanged without an initialised process group (world size 1)."""
    return global_argmin_of_rows(local_best(cost, payload, seed_offset), group)


def global_argmin_of_rows(row: torch.Tensor, group: Optional[dist.ProcessGroup] = None
                          ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """the exchange stage of ``global_argmin`` for callers that ran ``local_best`` themselves (e.g. captured behind the
    optimiser iterations in one hipGraph): packed rows [P, 2 + V] of this rank -> the winner over all ranks"""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return row[:, 0], row[:, 1].to(torch.int64), row[:, 2:]  # alone: the local best is the answer
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        flat = torch.empty(world * row.shape[0], row.shape[1], device=row.device, dtype=row.dtype)
        _all_gather_rows(flat, row, group)  # concatenated along dim 0
        gathered = flat.view(world, *row.shape)
    else:
        gathered = row.unsqueeze(0)
    costs = gathered[:, :, 0]  # [W, P]
    idxs = gathered[:, :, 1]
    # lexicographic (cost, global index) minimum -> identical on every rank
    best_cost = costs.min(dim=0).values
    is_best = costs == best_cost.unsqueeze(0)
    idx_masked = torch.where(is_best, idxs, torch.full_like(idxs, float("inf")))
    win_rank = idx_masked.argmin(dim=0)  # [P]
    P = row.shape[0]
    sel = gathered[win_rank, torch.arange(P, device=row.device)]
    return sel[:, 0], sel[:, 1].to(torch.int64), sel[:, 2:]


def local_topk(cost: torch.Tensor, payload: torch.Tensor, seed_offset: int, k: int) -> torch.Tensor:
    """cost[P, S_local], payload[P, S_local, V] -> packed [P, k, 2 + V] rows, best first.

    The reference ranks seeds with ``torch.topk(largest=False)`` (``solver/solver_ik.py:503-515``,
    ``util/tensor_util.py:178-179``), whose order among equal costs is unspecified; here ties are
    resolved towards the lowest seed index (stable sort), so the result does not depend on the
    sharding.  ``k`` is clamped to the local seed count (missing rows carry cost +inf, seed index -1 and a zero payload)."""
    P, S = cost.shape
    order = torch.sort(cost, dim=1, stable=True).indices[:, : min(k, S)]  # [P, k']
    rows = torch.zeros((P, k, 2 + payload.shape[-1]), device=cost.device, dtype=torch.float32)
    rows[:, :, 0] = float("inf")  # rows beyond the local seed count: cost +inf, seed index -1, zero payload
    rows[:, :, 1] = -1.0
    ar = torch.arange(P, device=cost.device).unsqueeze(1)
    kk = order.shape[1]
    rows[:, :kk, 0] = cost[ar, order]
    rows[:, :kk, 1] = (order + seed_offset).to(torch.float32)
    rows[:, :kk, 2:] = payload[ar, order]
    return rows


def global_topk(cost: torch.Tensor, payload: torch.Tensor, seed_offset: int, k: int,
                group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The k best (cost[P, k], global_seed_idx[P, k], payload[P, k, V]) over the seeds of ALL ranks,
    ordered by (cost, global seed index): every rank sends its local top-k (k rows per problem, one
    ``all_gather`` over RCCL), then ranks the W*k candidates identically."""
    rows = local_topk(cost, payload, seed_offset, k)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        world = dist.get_world_size(group)
        flat = torch.empty(world * rows.shape[0], *rows.shape[1:], device=rows.device, dtype=rows.dtype)
        _all_gather_rows(flat, rows, group)
        cand = flat.view(world, *rows.shape).permute(1, 0, 2, 3).reshape(rows.shape[0], world * k, -1)
    else:
        cand = rows
    # lexicographic (cost, global index): sort by index first, then stably by cost (padding rows, index -1,
    # have cost +inf and therefore end up last whatever their index)
    by_idx = torch.sort(cand[:, :, 1], dim=1, stable=True).indices
    cand = torch.gather(cand, 1, by_idx.unsqueeze(-1).expand_as(cand))
    by_cost = torch.sort(cand[:, :, 0], dim=1, stable=True).indices[:, :k]
    best = torch.gather(cand, 1, by_cost.unsqueeze(-1).expand(-1, -1, cand.shape[-1]))
    return best[:, :, 0], best[:, :, 1].to(torch.int64), best[:, :, 2:]



def gather_problem_winners(cost: torch.Tensor, payload: torch.Tensor, problem_offset: int, num_problems: int,
                           group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """PROBLEM shard (BASELINE config 5; SURVEY.md section 8e: "for the particle stage shard by problem"): this rank holds
    every seed of problems ``[problem_offset, problem_offset + P_local)``.  cost [P_local, S], payload [P_local, S, V] ->
    (best cost [num_problems], best seed index [num_problems], payload [num_problems, V]) on every rank: the local arg-min
    per problem, then ONE all-gather of the winners' packed rows (equal shards: num_problems divisible by the world size)."""
    row = local_best(cost, payload, 0)  # [P_local, 2 + V]
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        if row.shape[0] != num_problems:
            raise ValueError(f"alone in the process this rank must hold all {num_problems} problems, got {row.shape[0]}")
        return row[:, 0], row[:, 1].to(torch.int64), row[:, 2:]
    world = dist.get_world_size(group)
    if row.shape[0] * world != num_problems:
        raise ValueError(f"{num_problems} problems over {world} ranks need {num_problems // world} per rank, got {row.shape[0]}")
    flat = torch.empty(num_problems, row.shape[1], device=row.device, dtype=row.dtype)
    _all_gather_rows(flat, row, group)  # rank r's rows land at [r * P_local, (r + 1) * P_local): the problem order
    return flat[:, 0], flat[:, 1].to(torch.int64), flat[:, 2:]


def all_gather_problems(x: torch.Tensor, num_problems: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """problem shard: this rank's rows ``x [P_local, ...]`` -> the job's ``[num_problems, ...]`` on every rank (ranks own
    equal, contiguous problem ranges in rank order; identity alone in the process)"""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return x
    world = dist.get_world_size(group)
    if x.shape[0] * world != num_problems:
        raise ValueError(f"{num_problems} problems over {world} ranks need {num_problems // world} per rank, got {x.shape[0]}")
    flat = torch.empty(num_problems, *x.shape[1:], device=x.device, dtype=x.dtype)
    _all_gather_rows(flat, x, group)
    return flat
