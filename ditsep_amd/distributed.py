"""Data-parallel sharding of a batch of mixtures over the GPUs of one node.

Mixtures are independent (no term of the sampler couples batch items), so the path
shards by contiguous batch slices, every rank runs the whole path on its shard with a
full weight replica, and the ONLY collective is one gather of the separated waveforms
to rank 0 (RCCL over xGMI: each peer->root transfer rides its own direct link).
The reference's counterpart is a process pool with pickled results
(reference src/evaluate_latent.py:416-470, src/utils/processing_pool.py:90-166).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int, rank: int):
    """Contiguous, balanced [start, stop) of `rank`'s shard (first n_items % world ranks get one more)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def separate_sharded(separate_fn: Callable[[torch.Tensor], torch.Tensor], mix: torch.Tensor,
                     group: Optional[dist.ProcessGroup] = None, dst: int = 0):
    """Run `separate_fn` (mix_shard [b,1,L] -> wav [b,n,L]) on this rank's slice of `mix` (the full batch,
    identical on every rank) and gather the waveforms on `dst`.  Returns [B,n,L] on dst, None elsewhere."""
    if not dist.is_initialized():
        return separate_fn(mix)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = mix.shape[0]
    s, e = shard_bounds(B, world, rank)
    wav = separate_fn(mix[s:e]) if e > s else None
    sizes = [shard_bounds(B, world, r) for r in range(world)]
    max_b = max(b - a for a, b in sizes)
    # one padded gather (uneven shards only differ by one item)
    meta = torch.zeros(3, dtype=torch.long, device=mix.device)
    if wav is not None:
        meta[0], meta[1] = wav.shape[1], wav.shape[2]
    dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
    n, L = int(meta[0]), int(meta[1])
    buf = torch.zeros((max_b, n, L), dtype=torch.float32, device=mix.device)
    if wav is not None:
        buf[: e - s] = wav
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([o[: b - a] for o, (a, b) in zip(out, sizes)], dim=0)
