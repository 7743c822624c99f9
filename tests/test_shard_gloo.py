"""CPU: the N>1 exchange path (frame sharding + gather to rank 0) over gloo, world_size 2."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deeppointmap_amd.shard import gather_step_results, gather_to_root, shard_range


def test_shard_range_partitions_frames():
    for n, w in [(64, 8), (65, 8), (7, 2), (3, 4), (512, 3)]:
        blocks = [shard_range(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F = 3
    desc = torch.full((F, 131, 8), float(rank)) + torch.arange(F).view(F, 1, 1)
    table = torch.full((F, 56), float(10 * rank))
    d, t = gather_step_results(desc, table)
    if rank == 0:
        ok = tuple(d.shape) == (world * F, 131, 8) and tuple(t.shape) == (world * F, 56)
        for r in range(world):
            for f in range(F):
                ok &= bool((d[r * F + f] == r + f).all()) and bool((t[r * F + f] == 10 * r).all())
        q.put(ok)
    else:
        q.put(d is None and t is None)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_to_root_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)


def test_gather_is_identity_without_process_group():
    t = torch.arange(6.0).view(2, 3)
    assert gather_to_root(t) is t
