"""world_size-2 gloo tests of the N>1 host path: batch sharding, the single gather (the same
`ShardPlan.gather` / `separate_sharded` calls bench.py makes per step) and the rank launcher."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ditsep_amd import distributed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
        distributed.reset_plans()
        ok = True
        for rep in range(3):     # no plan passed: built on the first call, re-used afterwards (one gather per call)
            out = distributed.separate_sharded(_fake_separate, mix)
            assert len(distributed._PLANS) == 1
            if rank == 0:
                ok = ok and out is not None and torch.equal(out, _fake_separate(mix))
            else:
                assert out is None
        if rank == 0:
            results.put(bool(ok))
    finally:
        dist.destroy_process_group()


def _run(target, args, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = distributed.free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return q


@pytest.mark.parametrize("B", [4, 5, 1])
def test_sharded_separate_two_ranks(B):
    assert _run(_worker, (B,)).get(timeout=5) is True


def _bench_like_worker(rank, world, port, sizes, results):
    """What bench.py's step does: every rank owns its own (pre-sharded) inputs, the plan is built once,
    then several steps reuse it -- one gather per step, uneven shards included."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = sizes[rank]
        shard = torch.full((b, 1, 40), float(rank + 1)) + torch.arange(b, dtype=torch.float32).reshape(b, 1, 1)
        plan = distributed.ShardPlan(b, 2, 40, shard.device)
        assert plan.sizes == list(sizes) and plan.even == (len(set(sizes)) == 1)
        ok = True
        for step in range(3):
            out = distributed.separate_sharded(lambda m: _fake_separate(m) + step, shard, presharded=True, plan=plan)
            if rank == 0:
                want = torch.cat([_fake_separate(torch.full((s, 1, 40), float(r + 1))
                                                 + torch.arange(s, dtype=torch.float32).reshape(s, 1, 1)) + step
                                  for r, s in enumerate(sizes)], dim=0)
                ok = ok and out is not None and torch.equal(out, want)
            else:
                ok = ok and out is None
        # a wrong local shape is refused before the collective
        if b:
            with pytest.raises(ValueError):
                plan.gather(torch.zeros((b, 2, 39)))
            with pytest.raises(ValueError):          # no tensor / wrong dtype: refused before the collective too
                plan.gather(None)
            with pytest.raises(ValueError):
                plan.gather(torch.zeros((b, 2, 40), dtype=torch.float64))
        results.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [(3, 3), (3, 2), (2, 0)])
def test_presharded_plan_reuse_two_ranks(sizes):
    q = _run(_bench_like_worker, (sizes,))
    assert q.get(timeout=5) is True and q.get(timeout=5) is True


def test_unsharded_passthrough():
    mix = torch.randn(3, 1, 10)
    assert torch.equal(distributed.separate_sharded(_fake_separate, mix), _fake_separate(mix))


def test_launch_ranks_starts_one_process_per_rank(tmp_path):
    """`python bench.py --gpus N` without a launcher goes through distributed.launch_ranks: N fresh children under
    torch.distributed.run on 127.0.0.1.  Here: 2 CPU ranks (gloo) running the same sharded call."""
    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "from ditsep_amd import distributed\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "shard = torch.full((2, 1, 8), float(r))\n"
        "out = distributed.separate_sharded(lambda m: torch.cat([m, -m], 1), shard, presharded=True)\n"
        "if r == 0:\n"
        "    json.dump({'world': w, 'shape': list(out.shape), 'sum1': float(out[2:, 0].sum())}, open(sys.argv[1], 'w'))\n"
        "dist.destroy_process_group()\n")
    res = tmp_path / "res.json"
    rc = distributed.launch_ranks(str(child), [str(res)], 2)
    assert rc == 0
    got = json.load(open(res))
    assert got == {"world": 2, "shape": [4, 2, 8], "sum1": 16.0}


def test_bench_refuses_world_size_mismatch():
    """bench.py --gpus 2 under a 1-rank environment must exit non-zero, never fall back to one rank."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr
