"""BASELINE config 5 (infer_multiagents.py: agents + cloud, cross-agent loop closure) as far as one GPU allows: two AGENT
ranks and one CLOUD rank folded onto the test box's GPU, gloo in place of RCCL.  Each agent encodes its own slice of the
sequence, registers consecutive frames (odometry.py:103-127), keeps its trajectory and uploads every scan through the
rank-mapped Communicate_Module the way SlamSystem.step does (`UPLOAD_SCAN`, core.py:411-422).  The cloud -- CloudSystem's
loop (core.py:520-545) -- takes what arrives and runs the device side of its multi-agent loop closure
(loop_closure.py:166-258): loop_detection_forward of the new scan against the key-frames of the OTHER agent, then
registration_forward of the two agents' map tiles around the best pair.  Everything the cloud computed must equal what ONE
process computes from the same scans in the same arrival order."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FRAMES, N, TOKENS = 4, 8192, 128   # reduced_args: 128 descriptor tokens per scan


def _models(dev):
    from deeppointmap_amd.config import reduced_args
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.weights import init_procedural
    cfg = reduced_args()
    return init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev)


def _agent_scans(agent, enc, dec, dev):
    """what agent `agent` (1 or 2) uploads: per frame (token, descriptors, SE3_pred, odometry edge)"""
    from deeppointmap_amd import synthetic
    from deeppointmap_amd.registration import PoseTool, make_descriptors
    pts, pad = synthetic.frames(FRAMES, N, start=40 * agent)
    out, pose, prev = [], torch.eye(4), None
    for f in range(FRAMES):
        coor, fea, _ = enc(pts[f:f + 1], pad[f:f + 1])
        d = make_descriptors(coor, fea, 60.0)[0]
        edge = None
        if prev is not None:
            R, T, conf, rmse = dec.registration_forward(prev, d, num_sample=0.5)
            edge = PoseTool.SE3(R.cpu(), T.cpu()).inverse()
            pose = pose @ edge
        out.append(dict(token=100 * agent + f, agent=agent, timestep=f, key_points=d, SE3_pred=pose.clone(), odom=edge))
        prev = d
    return out


def _cloud_step(state, scan, dec, dev):
    """CloudSystem.step, device side: add the scan, then the multi-agent loop closure against the other agents' key-frames"""
    from deeppointmap_amd.registration import PoseTool
    store, scans = state
    store.put(scan["token"], scan["key_points"].to(dev))
    scans[scan["token"]] = scan
    others = [t for t, s in scans.items() if s["agent"] != scan["agent"]]
    if not others:
        return None
    src = torch.stack([scans[t]["key_points"].to(dev) for t in others])
    dst = scan["key_points"].to(dev).unsqueeze(0).repeat(len(others), 1, 1)
    prob = dec.loop_detection_forward(src, dst)
    best = others[int(torch.argmax(prob))]
    mine = [t for t, s in scans.items() if s["agent"] == scan["agent"]]
    theirs = [t for t, s in scans.items() if s["agent"] == scans[best]["agent"]]
    tile_a, _ = store.tile(theirs, [scans[t]["SE3_pred"] for t in theirs], scans[best]["SE3_pred"])
    tile_b, _ = store.tile(mine, [scans[t]["SE3_pred"] for t in mine], scan["SE3_pred"])
    R, T, conf, rmse = dec.registration_forward(tile_a, tile_b, num_sample=0.5)
    return dict(token=scan["token"], best=best, prob=prob.cpu(), R=R.cpu(), T=T.cpu(), rmse=rmse, n=int(conf.numel()))


def _worker(rank, world, port, q):
    try:
        _run(rank, world, port, q)
    except BaseException as e:  # noqa: BLE001 -- the parent must not wait for a queue entry that will never come
        q.put(("error", f"rank {rank}: {type(e).__name__}: {e}"))
        raise


def _run(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_grad_enabled(False)
    from deeppointmap_amd.comm import RankCommunicateModule
    from deeppointmap_amd.maptile import MapTileStore
    dev = torch.device("cuda:0")
    enc, dec = _models(dev)
    comm = RankCommunicateModule(device=dev)
    for m in range(world):
        comm.add_member(m)
    if rank == 0:   # the cloud (core.py:520-545)
        state, order, results, quit_ = (MapTileStore(dev, points=TOKENS), {}), [], [], set()
        while len(quit_) < world - 1:
            command, data = comm.fetch_message(0, block=True)
            if command == "UPLOAD_SCAN":
                scan = data["new_scan"]
                assert scan["key_points"].is_cuda        # a tensor that left a GPU arrives on the receiver's GPU
                order.append(scan["token"])
                r = _cloud_step(state, scan, dec, dev)
                if r is not None:
                    results.append(r)
            elif command == "AGENT_QUIT":
                quit_.add(data)
        for a in range(1, world):
            comm.send_message(caller=0, callee=a, command="QUIT", message=None)
        q.put((order, [{k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in r.items()} for r in results]))   # plain arrays: the sender may be gone before the parent reads
    else:           # an agent (core.py:360-423 with comm_module)
        for scan in _agent_scans(rank, enc, dec, dev):
            comm.send_message(caller=rank, callee=0, command="UPLOAD_SCAN",
                              message=dict(new_scan=scan, odometer_edge=scan["odom"], neighbor_edges=[]))
        comm.send_message(caller=rank, callee=0, command="AGENT_QUIT", message=rank)
        assert comm.fetch_message(rank, block=True) == ("QUIT", None)
    torch.cuda.synchronize()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_agents_and_a_cloud_equal_one_process():
    import torch.multiprocessing as mp
    from deeppointmap_amd.maptile import MapTileStore
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 90
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    order, got = q.get(timeout=240)
    assert order != "error", got
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(order) == sorted(100 * a + f for a in (1, 2) for f in range(FRAMES))
    for a in (1, 2):                                   # per agent in the order sent
        assert [t for t in order if t // 100 == a] == [100 * a + f for f in range(FRAMES)]
    # one process: the same scans (same kernels on the same frames: bit-identical descriptors and poses), the same order
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    enc, dec = _models(dev)
    scans = {s["token"]: s for a in (1, 2) for s in _agent_scans(a, enc, dec, dev)}
    state, want = (MapTileStore(dev, points=TOKENS), {}), []
    for t in order:
        r = _cloud_step(state, scans[t], dec, dev)
        if r is not None:
            want.append(r)
    assert len(got) == len(want) >= FRAMES               # every scan that found the other agent in the map was closed
    for g, w in zip(got, want):
        assert g["token"] == w["token"] and g["best"] == w["best"] and g["n"] == w["n"] and g["rmse"] == w["rmse"]
        for k in ("prob", "R", "T"):
            assert np.array_equal(g[k], w[k].numpy()), (g["token"], k)
