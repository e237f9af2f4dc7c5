"""N > 1 path on CPU: two gloo ranks, episodes sharded, no data-path collective (the per-rank compute is a stand-in:
the device kernels need a GPU; what is covered here is the sharding, the fenced max-over-ranks timing and the gather)."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from otter_amd.sharding import episode_shard, gather_actions, timed_region


def test_episode_shard_partitions():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [episode_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        episode_shard(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_episodes, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = episode_shard(n_episodes, rank, world)
        ids = torch.arange(lo, hi, dtype=torch.float32)
        calls = []

        def step():                                   # rank 1 is the slow one
            calls.append(1)
            time.sleep(0.02 * (1 + 2 * rank))

        elapsed = timed_region(step, 5, world=world)
        assert len(calls) == 5
        actions = ids[:, None, None].expand(hi - lo, 12, 10).contiguous() + 0.5 * rank
        full = gather_actions(actions, n_episodes, rank, world)
        torch.save({"elapsed": elapsed, "full": full}, os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_episodes", [7, 64])
def test_two_ranks_gloo(tmp_path, n_episodes):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_episodes, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    # every rank reports the same, slowest-rank time: rank 1 sleeps 5 * 60 ms
    assert res[0]["elapsed"] == res[1]["elapsed"]
    assert res[0]["elapsed"] >= 0.29
    assert res[1]["full"] is None
    full = res[0]["full"]
    assert full.shape == (n_episodes, 12, 10)
    split = episode_shard(n_episodes, 0, world)[1]
    expect = torch.arange(n_episodes, dtype=torch.float32)
    expect[split:] += 0.5
    assert torch.equal(full[:, 0, 0], expect) and torch.equal(full[:, 11, 9], expect)


def test_single_rank_passthrough():
    x = torch.zeros(3, 12, 10)
    assert gather_actions(x, 3, 0, 1) is x
    assert timed_region(lambda: None, 3) >= 0.0


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from otter_amd.trainer import allreduce_flat
        flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)          # rank r holds (r+1) * [0..999]
        allreduce_flat(flat, bucket_elems=300)                               # 4 buckets, the last one short
        torch.save(flat, os.path.join(out_dir, f"g{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_gradient_exchange_two_ranks_gloo(tmp_path):
    """Data-parallel gradient exchange of the training step: bucketed SUM all-reduce of the flat gradient buffer."""
    mp.spawn(_grad_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = torch.arange(1000, dtype=torch.float32) * 3
    for r in range(2):
        assert torch.equal(torch.load(tmp_path / f"g{r}.pt"), want)
