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
