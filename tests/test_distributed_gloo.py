"""world_size-2 gloo test of the batch sharding + single gather (the N>1 host path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ditsep_amd import distributed


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_separate(mix):
    # stands in for the GPU path: deterministic function of each mixture alone
    return torch.cat([mix * 2.0, mix * -3.0], dim=1)


def _worker(rank, world, port, B, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        mix = torch.randn((B, 1, 50), generator=g)
        out = distributed.separate_sharded(_fake_separate, mix)
        if rank == 0:
            ok = out is not None and torch.equal(out, _fake_separate(mix))
            results.put(bool(ok))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5, 1])
def test_sharded_separate_two_ranks(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_unsharded_passthrough():
    mix = torch.randn(3, 1, 10)
    assert torch.equal(distributed.separate_sharded(_fake_separate, mix), _fake_separate(mix))
