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
