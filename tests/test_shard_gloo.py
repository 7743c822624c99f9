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


def _halo_worker(rank, world, port, q):
    from deeppointmap_amd.shard import exchange_halo
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for step in range(3):  # the hand-over is one collective per rank and step: three in a row must stay matched
        desc = torch.full((131, 8), 100.0 * step + rank)
        pcd = torch.full((3, 40), 1000.0 * step + rank)
        got_d, got_p = exchange_halo(desc, pcd)
        prev = (rank - 1) % world
        ok &= tuple(got_d.shape) == (131, 8) and tuple(got_p.shape) == (3, 40)
        ok &= bool((got_d == 100.0 * step + prev).all()) and bool((got_p == 1000.0 * step + prev).all())
        got_d2, none = exchange_halo(desc, None)  # descriptors only (no scans: no information matrix)
        ok &= none is None and bool((got_d2 == 100.0 * step + prev).all())
    q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_ring_hands_the_last_frame_to_the_next_rank(world):
    """Block boundaries (reference odometry.py:103-127: every scan is registered against its predecessor): rank r gets
    the last frame of rank r-1, rank 0 the last frame of the whole window."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000 + world
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)
