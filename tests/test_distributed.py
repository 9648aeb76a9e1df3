"""Seed-sharded arg-min exchange, exercised with world_size 2 on the gloo backend (CPU)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from curobo_amd.distributed import global_argmin, local_best, shard_range


def test_shard_range_partitions_exactly():
    for n, w in [(256, 8), (10, 4), (7, 8), (512, 2)]:
        r = [shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1


def test_local_best_ties_pick_lowest_index():
    cost = torch.tensor([[2.0, 1.0, 1.0], [5.0, 5.0, 5.0]])
    payload = torch.arange(18, dtype=torch.float32).view(2, 3, 3)
    row = local_best(cost, payload, seed_offset=100)
    assert row[:, 1].tolist() == [101.0, 100.0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_seeds, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)  # identical global problem on every rank
    cost = torch.rand(3, total_seeds, generator=g)
    cost[1, 5] = cost[1, 40] = -1.0  # tie across ranks -> lowest global index wins
    payload = torch.rand(3, total_seeds, 6, generator=g)
    lo, hi = shard_range(total_seeds, rank, world)
    c, i, p = global_argmin(cost[:, lo:hi].contiguous(), payload[:, lo:hi].contiguous(), lo)
    ref_i = torch.argmin(cost, dim=1)
    assert torch.equal(i, ref_i), (rank, i, ref_i)
    assert i[1].item() == 5
    assert torch.equal(c, cost[torch.arange(3), ref_i])
    assert torch.equal(p, payload[torch.arange(3), ref_i])
    np.save(os.path.join(out_dir, f"r{rank}.npy"), torch.cat([c, i.float(), p.flatten()]).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_global_argmin_world_size_2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, 64, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    np.testing.assert_array_equal(a, b)  # every rank holds the identical winner


def test_global_topk_single_process_orders_by_cost_then_index():
    from curobo_amd.distributed import global_topk

    cost = torch.tensor([[3.0, 1.0, 1.0, 5.0, 0.5], [2.0, 2.0, 2.0, 2.0, 2.0]])
    payload = torch.arange(2 * 5 * 2, dtype=torch.float32).view(2, 5, 2)
    c, i, p = global_topk(cost, payload, seed_offset=10, k=3)
    assert c[0].tolist() == [0.5, 1.0, 1.0] and i[0].tolist() == [14, 11, 12]
    assert i[1].tolist() == [10, 11, 12]  # all equal: lowest indices, in order
    assert torch.equal(p[0, 0], payload[0, 4])
    # k larger than the seed count pads with +inf rows
    c2, i2, _ = global_topk(cost[:, :2].contiguous(), payload[:, :2].contiguous(), 0, k=3)
    assert torch.isinf(c2[:, 2]).all()


def _topk_worker(rank, world, port, total_seeds, k):
    from curobo_amd.distributed import global_topk

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1)
    cost = torch.rand(4, total_seeds, generator=g)
    cost[2, 3] = cost[2, 50] = cost[2, 20] = -1.0  # ties across ranks
    payload = torch.rand(4, total_seeds, 5, generator=g)
    lo, hi = shard_range(total_seeds, rank, world)
    c, i, p = global_topk(cost[:, lo:hi].contiguous(), payload[:, lo:hi].contiguous(), lo, k)
    ref = torch.sort(cost, dim=1, stable=True)
    assert torch.equal(i, ref.indices[:, :k]), (rank, i, ref.indices[:, :k])
    assert torch.equal(c, ref.values[:, :k])
    assert torch.equal(p, payload[torch.arange(4).unsqueeze(1), ref.indices[:, :k]])
    assert i[2, :3].tolist() == [3, 20, 50]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_global_topk_world_size_2_gloo():
    world, port = 2, _free_port()
    mp.spawn(_topk_worker, args=(world, port, 64, 5), nprocs=world, join=True)


# ---------------------------------------------------------------------------------------------------------------------
# Seed shards draw ONE global seed set ("W = 1 and W = 8 produce identical seed sets")
def _cpu_seed_solver(num_problems, num_seeds, seed_offset=0, global_num_seeds=None, **cfg_kw):
    from conftest import load_model
    from curobo_amd.robot.kinematics_params import KinematicsParams
    from curobo_amd.solver.seed_ik import SeedIKSolver, SeedIKSolverCfg

    kin = KinematicsParams.from_model(load_model("franka"), torch.device("cpu"))
    return SeedIKSolver(kin, num_problems, SeedIKSolverCfg(num_seeds=num_seeds, use_cuda_graph=False, **cfg_kw),
                        seed_offset=seed_offset, global_num_seeds=global_num_seeds)


def test_halton_seed_shards_are_slices_of_the_global_set():
    P, SG = 3, 24
    whole = _cpu_seed_solver(P, SG).generate_seeds()
    assert whole.shape == (P, SG, 7)
    for world in (2, 3, 8):
        parts = []
        for rank in range(world):
            lo, hi = shard_range(SG, rank, world)
            parts.append(_cpu_seed_solver(P, hi - lo, seed_offset=lo, global_num_seeds=SG).generate_seeds())
        assert torch.equal(torch.cat(parts, dim=1), whole), f"world size {world}"
    # the default joint position is the LAST seed of the global set, wherever that lands
    s = _cpu_seed_solver(P, 4, seed_offset=20, global_num_seeds=SG)
    assert torch.equal(s.generate_seeds()[:, -1], s.default_joint_position.view(1, -1).expand(P, -1))
    assert not torch.equal(_cpu_seed_solver(P, 4, seed_offset=0, global_num_seeds=SG).generate_seeds()[:, -1],
                           s.default_joint_position.view(1, -1).expand(P, -1))
    # caller seeds come first in the global set
    given = torch.rand(P, 2, 7)
    g0 = _cpu_seed_solver(P, 12, 0, SG).generate_seeds(given)
    g1 = _cpu_seed_solver(P, 12, 12, SG).generate_seeds(given)
    assert torch.equal(g0[:, :2], given) and torch.equal(torch.cat([g0, g1], 1), _cpu_seed_solver(P, SG).generate_seeds(given))
    with pytest.raises(ValueError, match="not inside the global seed"):
        _cpu_seed_solver(P, 8, seed_offset=20, global_num_seeds=SG)


def _lm_rank_worker(rank, world, port, out_dir):
    """every rank holds the LM results of its seed shard; the ranking over all shards (SeedIKSolver._rank_over_all_shards: one
    all-gather) must equal the single-process ranking of the whole seed set, on every rank"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, SG, k = 3, 20, 6
    g = torch.Generator().manual_seed(7)
    from conftest import load_model

    lim = torch.as_tensor(np.asarray(load_model("franka").joint_limits_position), dtype=torch.float32)
    q = 0.5 * (lim[0] + lim[1]) + 0.2 * (torch.rand(P, SG, 7, generator=g) - 0.5)  # inside the Franka limits
    pos_fail = (1, 7)  # one seed outside the tolerances: ranked last
    pos, ori = torch.rand(P, SG, generator=g) * 0.01, torch.rand(P, SG, generator=g) * 0.1
    pos[1, 4] = pos[1, 15] = 0.001
    ori[1, 4] = ori[1, 15] = 0.001  # a tie across the ranks: the lower global index wins
    pos[pos_fail] = 0.5
    lo, hi = shard_range(SG, rank, world)
    s = _cpu_seed_solver(P, hi - lo, seed_offset=lo, global_num_seeds=SG)
    assert s._sharded()
    s.q.view(P, hi - lo, 7).copy_(q[:, lo:hi])
    s.position_error.view(P, hi - lo).copy_(pos[:, lo:hi])
    s.orientation_error.view(P, hi - lo).copy_(ori[:, lo:hi])
    ok, sol, pe, oe = s._rank_over_all_shards(k, None)
    c = s.cfg
    good = (pos < c.position_tolerance) & (ori < c.orientation_tolerance)
    cost = pos + ori + 1e10 * (~good).float()
    order = torch.sort(cost, dim=1, stable=True).indices[:, :k]
    ar = torch.arange(P).unsqueeze(1)
    assert torch.equal(sol, q[ar, order]) and torch.equal(pe, pos[ar, order]) and torch.equal(ok, good[ar, order])
    assert order[1, :2].tolist() == [4, 15]
    # the exit test over all shards: a problem counts as solved when ANY rank holds a converged seed of it
    s.success.zero_()
    if rank == 1:
        s.success.view(P, hi - lo)[0, 0] = 1
        s.success.view(P, hi - lo)[2, 1] = 1
    s._stop_flag.zero_()
    s._global_batch_status(needed=3)
    assert int(s._stop_flag) == 0
    if rank == 0:
        s.success.view(P, hi - lo)[1, 3] = 1
    s._global_batch_status(needed=3)
    assert int(s._stop_flag) == 1
    np.save(os.path.join(out_dir, f"lm{rank}.npy"), sol.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_lm_seed_ranking_over_shards_world_size_2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_lm_rank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    np.testing.assert_array_equal(np.load(tmp_path / "lm0.npy"), np.load(tmp_path / "lm1.npy"))


# ---------------------------------------------------------------------------------------------------------------------
# PROBLEM shards (BASELINE config 5, the MPPI particle stage)
@pytest.mark.parametrize("noise_kw", [dict(), dict(fixed_samples=False, num_iters=3), dict(sample_ratio={"halton": 0.5, "stomp": 0.5}),
                                      dict(noise="torch")], ids=["halton_fixed", "halton_per_iteration", "halton_stomp", "torch"])
def test_mppi_problem_shards_sample_the_jobs_particles(noise_kw):
    """a problem's particle noise depends on its global index only: the shards' samples concatenate to the single-process ones
    (the reference's sample library, fixed or per iteration, with a STOMP share; and the torch.Generator noise)"""
    import functools

    from curobo_amd.optim.mppi import MPPI
    from curobo_amd.optim.mppi import MPPICfg as _Cfg

    MPPICfg = functools.partial(_Cfg, **noise_kw)
    PG, n_part, Ha, D = 6, 16, 5, 3
    lo_b, hi_b = -2.0 * torch.ones(D), 2.0 * torch.ones(D)
    mean = torch.rand(PG, Ha, D)
    whole = MPPI(MPPICfg(num_problems=PG, num_particles=n_part, null_act_frac=0.125), lambda a: a.sum(-1), Ha, D, (lo_b, hi_b), "cpu")
    whole.mean.copy_(mean)
    want = [whole.sample_actions().clone() for _ in range(2)]  # two iterations: the generator state advances alike
    if noise_kw.get("fixed_samples", True) and noise_kw.get("noise") != "torch":
        assert torch.equal(want[0], want[1])  # fixed samples: one noise set for every iteration (reference default)
    else:
        assert not torch.equal(want[0], want[1])
    for world in (2, 3):
        parts = [[], []]
        for rank in range(world):
            lo, hi = shard_range(PG, rank, world)
            m = MPPI(MPPICfg(num_problems=hi - lo, num_particles=n_part, null_act_frac=0.125), lambda a: a.sum(-1), Ha, D, (lo_b, hi_b),
                     "cpu", problem_offset=lo, global_num_problems=PG)
            m.mean.copy_(mean[lo:hi])
            for it in range(2):
                parts[it].append(m.sample_actions().clone())
        for it in range(2):
            assert torch.equal(torch.cat(parts[it], 0), want[it]), (world, it)
    with pytest.raises(ValueError, match="problem shard"):
        MPPI(MPPICfg(num_problems=4), lambda a: a, Ha, D, (lo_b, hi_b), "cpu", problem_offset=4, global_num_problems=6)


def _problem_worker(rank, world, port):
    from curobo_amd.distributed import all_gather_problems, gather_problem_winners

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    PG, S, V = 8, 10, 4
    g = torch.Generator().manual_seed(3)
    cost, payload = torch.rand(PG, S, generator=g), torch.rand(PG, S, V, generator=g)
    cost[5, 2] = cost[5, 7] = -1.0  # a tie inside a problem: the lower seed index
    lo, hi = shard_range(PG, rank, world)
    c, i, p = gather_problem_winners(cost[lo:hi].contiguous(), payload[lo:hi].contiguous(), lo, PG)
    ref = cost.argmin(1)
    assert torch.equal(i, ref) and i[5].item() == 2
    assert torch.equal(c, cost[torch.arange(PG), ref]) and torch.equal(p, payload[torch.arange(PG), ref])
    full = all_gather_problems(payload[lo:hi].contiguous(), PG)
    assert torch.equal(full, payload)
    try:
        gather_problem_winners(cost[:3].contiguous(), payload[:3].contiguous(), 0, PG)
        raise AssertionError("unequal problem shards must be refused")
    except ValueError:
        pass
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_problem_shard_exchange_world_size_2_gloo():
    mp.spawn(_problem_worker, args=(2, _free_port()), nprocs=2, join=True)


# ---------------------------------------------------------------------------------------------------------------------
def _random_exchange_worker(rank, world, port, seed):
    """random costs with many ties (quantised to a few values), uneven problem counts and payload widths: the arg-min and the
    top-k over the ranks must be the stable-sort answer over the global seed set, on every rank"""
    from curobo_amd.distributed import global_topk

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(seed)  # the same global problem on every rank
    for _ in range(4):
        P = int(torch.randint(1, 9, (1,), generator=g))
        per = int(torch.randint(1, 7, (1,), generator=g))
        total, V = per * world, int(torch.randint(1, 9, (1,), generator=g))
        levels = int(torch.randint(2, 6, (1,), generator=g))
        cost = torch.randint(0, levels, (P, total), generator=g).float() * 0.25  # a handful of distinct values: ties everywhere
        payload = torch.rand(P, total, V, generator=g)
        lo, hi = shard_range(total, rank, world)
        c, i, p = global_argmin(cost[:, lo:hi].contiguous(), payload[:, lo:hi].contiguous(), lo)
        ref = torch.sort(cost, dim=1, stable=True)
        assert torch.equal(i, ref.indices[:, 0]) and torch.equal(c, ref.values[:, 0]), (rank, world, i, ref.indices[:, 0])
        assert torch.equal(p, payload[torch.arange(P), ref.indices[:, 0]])
        k = int(torch.randint(1, total + 1, (1,), generator=g))
        c, i, p = global_topk(cost[:, lo:hi].contiguous(), payload[:, lo:hi].contiguous(), lo, k)
        assert torch.equal(i, ref.indices[:, :k]) and torch.equal(c, ref.values[:, :k]), (rank, world, k)
        assert torch.equal(p, payload[torch.arange(P).unsqueeze(1), ref.indices[:, :k]])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [3, 4, 8])
def test_random_exchanges_with_ties_world_sizes_3_4_8_gloo(world):
    """the exchange of the path at the world sizes of BASELINE's configs (4 and 8 ranks; 3 for an uneven one), gloo on the CPU"""
    mp.spawn(_random_exchange_worker, args=(world, _free_port(), 100 + world), nprocs=world, join=True)
