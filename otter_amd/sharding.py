"""Episode sharding across GPUs: one process per GPU, episodes are independent, no collective on the data path.

The reference serves a single episode per process (``otter/policy/otter_interface.py:142-185``); batches of
independent episodes shard by episode id, weights replicated, each rank keeping the rollout state of the
episodes it owns.  The only exchanges are host-side: the barrier / max-over-ranks of a timed region and, when a
caller wants all results in one place, a gather of the (episodes, horizon, action_dim) action chunks.
"""
from __future__ import annotations

import time
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def episode_shard(n_episodes: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of the episodes rank owns: contiguous, sizes differ by at most one, earlier ranks get the extra."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if n_episodes < 0:
        raise ValueError("negative episode count")
    base, extra = divmod(n_episodes, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _active(world: int) -> bool:
    return world > 1 and dist.is_available() and dist.is_initialized()


def timed_region(step: Callable[[], object], steps: int, *, world: int = 1, sync: Optional[Callable[[], None]] = None,
                 device: Optional[torch.device] = None) -> float:
    """Run ``step`` exactly ``steps`` times between two (device sync + barrier) fences and return the elapsed
    seconds, MAX over ranks.  ``sync`` drains the rank's device queue (``torch.cuda.synchronize``)."""
    sync = sync or (lambda: None)
    on = _active(world)
    sync()
    if on:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if on:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def gather_actions(local: Tensor, n_episodes: int, rank: int, world: int) -> Optional[Tensor]:
    """Collect every rank's (n_local, horizon, action_dim) action chunks on rank 0 in episode order
    (shards as :func:`episode_shard`).  Returns the (n_episodes, horizon, action_dim) tensor on rank 0, else None."""
    lo, hi = episode_shard(n_episodes, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} episodes, its shard has {hi - lo}")
    if not _active(world):
        return local
    biggest = -(-n_episodes // world)
    padded = local.new_zeros((biggest,) + tuple(local.shape[1:]))
    padded[: hi - lo] = local
    parts = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, parts, dst=0)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        a, b = episode_shard(n_episodes, r, world)
        out.append(parts[r][: b - a])
    return torch.cat(out, 0)
