"""Seed-sharded L-BFGS on several HIP streams of one GPU.

The optimisation problems (seeds) of an L-BFGS batch are independent, and one iteration is a
dependent chain  rollout -> line search + two-loop -> next rollout  in which the optimiser-side
kernel (a few workgroups, latency bound) leaves the chip almost idle.  Splitting the seeds into
``n_shards`` optimisers that run on their own streams lets the optimiser kernel of one shard
execute while the rollout workgroups of the other shards fill the CUs; the shards share nothing,
so the iterates are exactly those of one big batch (same arithmetic per seed).  All shards'
``inner_iters`` iterations are captured into ONE hipGraph with a fork/join over the streams.

This is the single-GPU counterpart of the seed sharding across ranks in
``curobo_amd/distributed.py`` (reference: one batch on one stream,
``optim/components/gradient_opt_core.py:255-480`` under ``util/cuda_graph_util.py:144-180``).
"""

from __future__ import annotations

import dataclasses
from typing import Callable, List, Optional, Tuple

import torch

from ..util.stream_scope import forked_stream
from .lbfgs import LBFGSOpt, LBFGSOptCfg


class PipelinedLBFGS:
    """``rollout_factory(batch[, shard_index]) -> cost_and_gradient`` builds one rollout per shard
    (batch = seeds of the shard x line-search candidates).  The interface follows :class:`LBFGSOpt`."""

    def __init__(self, cfg: LBFGSOptCfg, rollout_factory: Callable[[int], Callable], action_horizon: int,
                 action_dim: int, action_bounds: Tuple[torch.Tensor, torch.Tensor], device, n_shards: int = 2,
                 use_cuda_graph: bool = True):
        if cfg.num_problems % n_shards != 0:
            raise ValueError(f"num_problems ({cfg.num_problems}) must be a multiple of n_shards ({n_shards})")
        self.cfg, self.device, self.n_shards = cfg, device, n_shards
        self.use_cuda_graph = use_cuda_graph
        self.shard_problems = cfg.num_problems // n_shards
        sub = dataclasses.replace(cfg, num_problems=self.shard_problems)
        nls = len(cfg.line_search_scale)
        import inspect

        takes_index = len(inspect.signature(rollout_factory).parameters) >= 2  # factory(batch, shard_index)
        self.opts: List[LBFGSOpt] = [
            LBFGSOpt(sub, rollout_factory(self.shard_problems * nls, k) if takes_index else rollout_factory(self.shard_problems * nls),
                     action_horizon, action_dim, action_bounds, device, use_cuda_graph=False)
            for k in range(n_shards)]
        for o in self.opts:
            o.overlapped = n_shards > 1
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n_shards)]
        self.action_horizon, self.action_dim = action_horizon, action_dim
        self._graph: Optional[torch.cuda.CUDAGraph] = None

    # ------------------------------------------------------------------ fork / join
    def _forked(self, fn: Callable[[LBFGSOpt], None]) -> None:
        cur = torch.cuda.current_stream(self.device)
        for opt, s in zip(self.opts, self.streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s), forked_stream():  # (a rollout with a side stream of its own keeps to the shard's stream)
                fn(opt)
        for s in self.streams:
            cur.wait_stream(s)

    def reinitialize(self, seed: torch.Tensor) -> None:
        parts = seed.reshape(self.n_shards, self.shard_problems, -1)
        for opt, part in zip(self.opts, parts):
            opt.reinitialize(part)

    def step(self) -> None:
        """one iteration of every shard (eager launches)"""
        self._forked(lambda o: o._opt_step())

    def capture(self) -> None:
        saved = [[t.clone() for t in o._state_tensors()] for o in self.opts]
        self.step()  # warm-up outside the capture (lazy module loads, workspace set-up)
        torch.cuda.synchronize(self.device)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._forked(lambda o: o._opt_iters())
        for o, sv in zip(self.opts, saved):
            for t, s in zip(o._state_tensors(), sv):
                t.copy_(s)
        torch.cuda.synchronize(self.device)

    def make_graph(self, n_iters: int, after: Optional[Callable[[], None]] = None) -> "torch.cuda.CUDAGraph":
        """a hipGraph of ``n_iters`` iterations of every shard (state is restored after the capture); ``after`` is captured
        behind the join of the shards (e.g. the local stage of an arg-min exchange: one replay = iterations + reduction)"""
        saved = [[t.clone() for t in o._state_tensors()] for o in self.opts]
        self.step()
        if after is not None:
            after()  # warm-up outside the capture
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._forked(lambda o: [o._opt_step() for _ in range(n_iters)])
            if after is not None:
                after()
        for o, sv in zip(self.opts, saved):
            for t, s in zip(o._state_tensors(), sv):
                t.copy_(s)
        torch.cuda.synchronize(self.device)
        return g

    def run_inner(self) -> None:
        if self.use_cuda_graph:
            if self._graph is None:
                self.capture()
            self._graph.replay()
        else:
            self._forked(lambda o: o._opt_iters())

    def optimize(self, seed: torch.Tensor) -> torch.Tensor:
        """``cfg.fixed_iters=False``: stop after the first block of ``inner_iters`` iterations at whose end more than
        ``cfg.converged_ratio`` of ALL problems (over the shards) have converged (reference BestTracker.check_convergence,
        optim/components/best_tracker.py:109-118; one device -> host read per block, as in ``LBFGSOpt.optimize``)."""
        self.reinitialize(seed)
        self.iterations_run = 0
        for _ in range(max(1, self.cfg.num_iters // self.cfg.inner_iters)):
            self.run_inner()
            self.iterations_run += self.cfg.inner_iters
            if not self.cfg.fixed_iters and self._enough_converged():
                break
        return self.best_action.view(self.cfg.num_problems, self.action_horizon, self.action_dim)

    def _enough_converged(self) -> bool:
        n = sum(torch.count_nonzero(o.converged).to(torch.float32) for o in self.opts).reshape(1)
        total = float(sum(o.converged.numel() for o in self.opts))
        if any(o.rank_sharded for o in self.opts):
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                from ..distributed import all_reduce_sum

                both = all_reduce_sum(torch.cat([n, torch.tensor([total], device=n.device)]))
                n, total = both[:1], float(both[1])
        return float(n.item()) > total * self.cfg.converged_ratio

    # ------------------------------------------------------------------ results (seed order of the input)
    @property
    def best_cost(self) -> torch.Tensor:
        return torch.cat([o.best_cost for o in self.opts])

    @property
    def best_action(self) -> torch.Tensor:
        return torch.cat([o.best_action for o in self.opts])

    @property
    def best_iteration(self) -> torch.Tensor:
        return torch.cat([o.best_iteration for o in self.opts])

    @property
    def x_set(self) -> torch.Tensor:
        """line-search candidates of the next iteration, [num_problems, n_linesearch, opt_dim]"""
        return torch.cat([o.x_set for o in self.opts])
