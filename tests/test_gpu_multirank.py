"""Two ranks folded onto the one GPU of the test box (gloo rendezvous on 127.0.0.1, the DPM_BENCH_BACKEND=gloo dry-run
path of bench.py): the HIP hot path runs under a process group, block-boundary edges come from the neighbour rank's
last frame (shard.exchange_halo), and rank 0 receives exactly the descriptors and edge tables that ONE process
produces for the same frames in sequence."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

F, N = 3, 8192


def _hot(dev):
    from deeppointmap_amd.config import reduced_args
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.pipeline import HotPath
    from deeppointmap_amd.weights import init_procedural
    cfg = reduced_args()
    hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
    hot.chain = True
    return hot


def _block(rank, step, world, dev):
    """frames of rank `rank` in step `step`: the global sequence is cut into steps of world * F frames"""
    from deeppointmap_amd import synthetic
    pts, pad = synthetic.frames(F, N, start=(step * world + rank) * F)
    return pts.to(dev), pad.to(dev), (pts * 60.0).contiguous().to(dev)


def _worker(rank, world, port, q, pipelined):
    import torch.distributed as dist
    from deeppointmap_amd.shard import gather_step_results
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    hot = _hot(dev)
    got = []
    for step in range(2):
        pts, pad, pcd = _block(rank, step, world, dev)
        if pipelined:
            done = hot.submit(pts, pad, pcd)
            outs = [done] if done is not None else []
        else:
            desc, _, table = hot.step(pts, pad, pcd, materialize=False)
            outs = [(desc, table)]
        for d, t in outs:
            gd, gt = gather_step_results(d.contiguous(), t)
            if rank == 0:
                got.append((gd.cpu(), gt.cpu()))
    if pipelined:
        for d, t in hot.flush():
            gd, gt = gather_step_results(d.contiguous(), t)
            if rank == 0:
                got.append((gd.cpu(), gt.cpu()))
    torch.cuda.synchronize()
    if rank == 0:
        q.put([(d.numpy(), t.numpy()) for d, t in got])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pipelined", [False, True])
def test_two_ranks_equal_one_process_in_sequence(pipelined):
    import numpy as np
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200 + (50 if pipelined else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, pipelined)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # one process, the same 4 blocks in sequence order (step 0: rank 0, rank 1; step 1: rank 0, rank 1)
    dev = torch.device("cuda:0")
    hot = _hot(dev)
    want = []
    for step in range(2):
        blocks = []
        for rank in range(world):
            desc, _, table = hot.step(*_block(rank, step, world, dev), materialize=False)
            blocks.append((desc.cpu().numpy().copy(), table.cpu().numpy().copy()))
        want.append((np.concatenate([b[0] for b in blocks]), np.concatenate([b[1] for b in blocks])))
    assert len(got) == 2
    for s, ((gd, gt), (wd, wt)) in enumerate(zip(got, want)):
        assert gd.shape[:2] == (world * F, 131) and gt.shape == (world * F, 56)
        assert np.array_equal(gd, wd), s                       # descriptors: the very same kernels on the same frames
        # edges: rank 1's first edge has rank 0's last frame as its source (halo), and from step 1 on rank 0's first
        # edge has rank 1's last frame of the step before: all of it equals the single process walking the sequence
        np.testing.assert_allclose(gt, wt, rtol=0, atol=0, err_msg=f"step {s}")
    # and the boundary edge is a real one: not the ring edge a lone batch would have produced
    lone = _hot(dev)
    lone.chain = False
    _, _, ring_table = lone.step(*_block(1, 0, world, dev), materialize=False)
    assert not np.allclose(ring_table.cpu().numpy()[0, :12], want[0][1][F, :12])


def test_bench_two_ranks_dry_run():
    """`bench.py --gpus 2` exactly as the driver launches it (python -m torch.distributed.run, one process per rank), with the
    ranks folded onto this box's one GPU and gloo in place of RCCL (--backend gloo): the N > 1 code path -- chain mode,
    halo hand-over, the gather of every step, the rank-0 consumer -- runs end to end and prints the contract's JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29950 + os.getpid() % 40), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--cpu-frames", "0", "--frames", "8", "--points", "16384", "--backend", "gloo"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["unit"] == "frames/s" and line["higher_is_better"] is True
    assert line["config"]["frames_per_gpu_per_step"] == 8 and "DRY RUN over gloo" in line["config"]["parallelism"]
    assert line["rank0_consumer"]["synthetic_rows"] is True
    assert line["rank0_serial_ms"] > 0 and line["value_with_rank0_consumer"] > 0
    assert "roofline" in line and line["roofline"]["bound"] == "hbm"


def test_bench_two_ranks_started_without_a_launcher():
    """`python bench.py --gpus 2 ...` -- the N = 1 command form with N = 2, no torch.distributed.run around it: bench.py becomes the
    launcher itself (bench.self_launch_command) and rank 0's line reports a world of two."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-frames", "0",
           "--frames", "8", "--points", "16384", "--backend", "gloo", "--no-extras"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "no launcher in the environment" in out.stderr
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"]["world_size"] == 2 and line["value"] > 0
