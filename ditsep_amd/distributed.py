"""Data-parallel sharding of a batch of mixtures over the GPUs of one node.

Mixtures are independent (no term of the sampler couples batch items), so the path
shards by contiguous batch slices, every rank runs the whole path on its shard with a
full weight replica, and the ONLY collective on the data path is one gather of the
separated waveforms to rank 0 (RCCL over xGMI: each peer->root transfer rides its own
direct link).  The reference's counterpart is a process pool with pickled results
(reference src/evaluate_latent.py:416-470, src/utils/processing_pool.py:90-166).

`bench.py` and `separate_sharded` both go through `ShardPlan.gather` -- the function the
world-size-2 gloo test drives is the one the benchmark times.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int, rank: int):
    """Contiguous, balanced [start, stop) of `rank`'s shard (first n_items % world ranks get one more)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class ShardPlan:
    """Who holds how many mixtures, and the buffers of the single gather.  Built once per (shape, group) --
    building exchanges the shard sizes (one tiny all_gather, setup only); `gather` is then exactly one
    collective per call with no metadata traffic."""

    def __init__(self, local_items: int, n_src: int, length: int, device, group=None, dst: int = 0,
                 sizes: Optional[Sequence[int]] = None):
        self.group, self.dst = group, dst
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if sizes is None:
            mine = torch.tensor([int(local_items)], dtype=torch.long, device=device)
            every = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(every, mine, group=group)
            sizes = [int(t.item()) for t in every]
        self.sizes: List[int] = [int(s) for s in sizes]
        if self.sizes[self.rank] != int(local_items):
            raise ValueError(f"rank {self.rank}: plan says {self.sizes[self.rank]} items, shard has {local_items}")
        self.n_src, self.length, self.device = int(n_src), int(length), device
        self.send_device = torch.empty(0, device=device).device      # normalised ("cuda" -> cuda:<current>)
        self.max_items = max(self.sizes)
        self.even = min(self.sizes) == self.max_items
        shape = (self.max_items, self.n_src, self.length)
        # uneven shards are padded to the largest (they differ by at most one item under shard_bounds)
        self.send = None if self.even else torch.zeros(shape, dtype=torch.float32, device=device)
        self.recv = ([torch.empty(shape, dtype=torch.float32, device=device) for _ in range(self.world)]
                     if self.rank == dst else None)

    def gather(self, wav_local: Optional[torch.Tensor]):
        """The one data-path collective: every rank's [b_r, n, L] waveforms -> list of per-rank tensors on
        dst (views into the plan's receive buffers, valid until the next gather), None elsewhere."""
        b = self.sizes[self.rank]
        if b:
            # checked BEFORE the collective: a rank that raises after its peers entered dist.gather hangs them
            if wav_local is None:
                raise ValueError(f"rank {self.rank} owns {b} mixtures but passed no waveforms")
            if tuple(wav_local.shape) != (b, self.n_src, self.length):
                raise ValueError(f"expected {(b, self.n_src, self.length)}, got {tuple(wav_local.shape)}")
            if wav_local.dtype != torch.float32 or wav_local.device != self.send_device:
                raise ValueError(f"waveforms must be float32 on {self.send_device} (the plan's buffers), got "
                                 f"{wav_local.dtype} on {wav_local.device}")
        if self.even:
            send = wav_local.contiguous()
        else:
            send = self.send
            if b:
                send[:b].copy_(wav_local)
        dist.gather(send, self.recv, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        return [buf[:s] for buf, s in zip(self.recv, self.sizes)]


_PLANS: dict = {}      # separate_sharded's plans when the caller passes none (cleared by reset_plans)


def reset_plans():
    """Forget the cached ShardPlans (call after destroying / re-creating the process group)."""
    _PLANS.clear()


def separate_sharded(separate_fn: Callable[[torch.Tensor], torch.Tensor], mix: torch.Tensor,
                     group: Optional[dist.ProcessGroup] = None, dst: int = 0, *, presharded: bool = False,
                     plan: Optional[ShardPlan] = None):
    """Run `separate_fn` (mix_shard [b,1,L] -> wav [b,n,L]) on this rank's shard and gather the waveforms on
    `dst`.  `mix` is the full batch, identical on every rank (sliced here by shard_bounds), or -- with
    `presharded` -- already this rank's own shard (weak scaling: bench.py).  Returns [B,n,L] on dst (rank
    order = batch order), None elsewhere.  Pass a `plan` to reuse buffers across calls."""
    if not dist.is_initialized():
        return separate_fn(mix)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if presharded:
        shard = mix
    else:
        s, e = shard_bounds(mix.shape[0], world, rank)
        shard = mix[s:e]
    wav = separate_fn(shard) if shard.shape[0] > 0 else None
    if plan is None:
        # No plan passed: build one ONCE per (shard size, device, group, dst) and keep it -- the geometry exchange
        # below (and ShardPlan's own size all_gather when presharded) are set-up collectives, not part of the data
        # path; later calls with the same key issue the gather only.  The hot path (bench.py) passes its plan.
        key = (shard.shape[0], mix.shape[0], bool(presharded), str(mix.device), id(group), dst)
        plan = _PLANS.get(key)
        if plan is None:
            # output geometry from whoever has items (an empty shard cannot know n, L)
            meta = torch.zeros(2, dtype=torch.long, device=mix.device)
            if wav is not None:
                meta[0], meta[1] = wav.shape[1], wav.shape[2]
            dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
            sizes = None if presharded else [b - a for a, b in (shard_bounds(mix.shape[0], world, r)
                                                                 for r in range(world))]
            plan = ShardPlan(shard.shape[0], int(meta[0]), int(meta[1]), mix.device, group, dst, sizes=sizes)
            _PLANS[key] = plan
    parts = plan.gather(wav)
    if parts is None:
        return None
    return torch.cat(parts, dim=0)


# ------------------------------------------------------------------ launching one process per GPU
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(script: str, argv: Sequence[str], n_ranks: int) -> int:
    """Start `script argv` as n_ranks fresh processes, one per GPU, through torch.distributed.run on
    127.0.0.1 (the same command line the round driver uses) and return its exit code.  Must be called
    BEFORE the calling process has touched the GPU: ranks are children, nothing is re-exec'd."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: required by RCCL on this driver
    return subprocess.run(cmd, env=env).returncode
