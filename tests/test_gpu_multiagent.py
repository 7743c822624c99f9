"""BASELINE config 5 (infer_multiagents.py: agents + cloud, cross-agent loop closure) as far as one GPU allows: two AGENT
ranks and one CLOUD rank folded onto the test box's GPU, gloo in place of RCCL.  Each agent is a
deeppointmap_amd.system.AgentSystem: it runs the reference's step on its own scans (core.py:360-423) and uploads every
accepted key-frame with its edges through the rank-mapped Communicate_Module (`UPLOAD_SCAN`, core.py:409-422).  The cloud is a
CloudSystem (core.py:451-546): every upload joins its graph at the end of its odometry edge, then the multi-agent loop
closure runs -- loop_detection_forward against the OTHER agent's key-frames, map-to-map registration of the two
neighbourhood tiles, verification, optimisation over both agents' key-frames, merged coordinate systems.  Everything the
cloud ends up with must equal what ONE process computes from the same uploads in the same arrival order."""
import os
from queue import Queue

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FRAMES, N = 4, 8192              # reduced_args: 128 descriptor tokens per scan

# thresholds under which, with procedural weights and synthetic scans, every scan is a key-frame and every loop proposal is
# believed (the decisions are exercised by tests/test_gpu_consumer.py on the recorded run; here the traffic is)
SLAM = dict(edge_confidence_drop=0.0, edge_rmse_drop=1e9, key_frame_distance=0.0, loop_detection_gap=0,
            loop_detection_transaction_gap=0.0, loop_detection_pred_distance=1e9, loop_detection_rotation_min=0.0,
            loop_detection_translation_min=0.0, loop_detection_prob_acpt_threshold=0.0,
            loop_detection_confidence_acpt_threshold=0.0)


def _args():
    from deeppointmap_amd.config import Cfg, reduced_args
    a = reduced_args()
    a.device, a.slam_system = "cuda:0", Cfg(SLAM)
    return a


def _models(dev):
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.weights import init_procedural
    cfg = _args()
    return init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev)


def _loader(agent):
    from deeppointmap_amd import synthetic
    pts, pad = synthetic.frames(FRAMES, N, start=40 * agent)
    return [[pts[f:f + 1], torch.eye(3).unsqueeze(0), torch.zeros(1, 3, 1), pad[f:f + 1], None] for f in range(FRAMES)]


class LocalComm:
    """the reference's Communicate_Module (system/modules/utils.py:116-154): a dict of queues shared by threads"""

    def __init__(self):
        self.queues, self.sent = {}, []

    def add_member(self, m):
        self.queues.setdefault(m, Queue())

    def send_message(self, caller, callee, command, message):
        self.sent.append((caller, callee, command, message))
        self.queues[callee].put((command, message))

    def fetch_message(self, m, block=True):
        return self.queues[m].get() if block or not self.queues[m].empty() else ("NO_OP", None)


def _cloud_summary(cloud):
    b = cloud.backend
    toks = sorted(b.poses)
    return dict(arrivals=list(cloud.arrivals), tokens=toks, poses=torch.stack([b.poses[t] for t in toks]).numpy(),
                desc=torch.stack([b.desc[t].cpu() for t in toks]).numpy(),
                coor=[b.coor[t] for t in toks], edges=[(a, c, e["type"]) for (a, c), e in b.edges.items()],
                edge_SE3=np.stack([e["SE3"].numpy() for e in b.edges.values()]), stats=dict(b.stats))


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
    from deeppointmap_amd.system import AgentSystem, CloudSystem
    dev = torch.device("cuda:0")
    enc, dec = _models(dev)
    comm = RankCommunicateModule(device=dev)
    for m in range(world):
        comm.add_member(m)
    if rank == 0:   # the cloud (core.py:520-545)
        cloud = CloudSystem(_args(), enc, dec, comm_module=comm, device=dev)
        cloud.start(expected_agents=world - 1)
        cloud.wait()
        assert all(cloud.backend.desc[t].is_cuda for t in cloud.arrivals)   # a tensor that left a GPU arrives on the receiver's GPU
        for a in range(1, world):
            comm.send_message(caller=0, callee=a, command="QUIT", message=None)
        q.put(("ok", _cloud_summary(cloud)))            # plain arrays: the sender may be gone before the parent reads
    else:           # an agent (core.py:426-448)
        agent = AgentSystem(_args(), enc, dec, system_id=rank, comm_module=comm, device=dev)
        agent.start(_loader(rank))
        agent.wait()
        comm.send_message(caller=rank, callee=0, command="AGENT_QUIT", message=rank)
        assert comm.fetch_message(rank, block=True) == ("QUIT", None)
    torch.cuda.synchronize()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_agents_and_a_cloud_equal_one_process():
    import torch.multiprocessing as mp
    from deeppointmap_amd.system import AgentSystem, CloudSystem
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 90
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    status, got = q.get(timeout=240)
    assert status == "ok", got
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    order = got["arrivals"]
    tok = lambda a, f: (a << 16) + f
    assert sorted(order) == sorted(tok(a, f) for a in (1, 2) for f in range(FRAMES))
    for a in (1, 2):                                   # per agent in the order sent
        assert [t for t in order if t >> 16 == a] == [tok(a, f) for f in range(FRAMES)]
    # one process: the same agents (same kernels on the same scans: bit-identical uploads), the cloud fed in the same order
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    enc, dec = _models(dev)
    comm = LocalComm()
    comm.add_member(0)
    for a in (1, 2):
        agent = AgentSystem(_args(), enc, dec, system_id=a, comm_module=comm, device=dev)
        agent.start(_loader(a))
        agent.wait()
    uploads = {m["new_scan"]["token"]: m for _, _, c, m in comm.sent if c == "UPLOAD_SCAN"}
    assert len(uploads) == 2 * FRAMES
    cloud = CloudSystem(_args(), enc, dec, comm_module=LocalComm(), device=dev)
    for t in order:
        m = uploads[t]
        cloud.step(scan_pack=m["new_scan"], odom_edge=m["odometer_edge"], neighbor_edges=m["neighbor_edges"])
    want = _cloud_summary(cloud)
    assert got["tokens"] == want["tokens"] and got["edges"] == want["edges"] and got["stats"] == want["stats"]
    assert got["coor"] == want["coor"]
    assert np.array_equal(got["desc"], want["desc"]), f"the agents' descriptors differ by {np.abs(got['desc'] - want['desc']).max():.3e}"
    dp, de = float(np.abs(got["poses"] - want["poses"]).max()), float(np.abs(got["edge_SE3"] - want["edge_SE3"]).max())
    assert dp == 0.0 and de == 0.0, f"cloud of ranks vs one process: poses differ by {dp:.3e}, edge transforms by {de:.3e}"
    # the traffic did what config 5 is about: loops between the agents were closed, the optimiser ran over both agents'
    # key-frames and their coordinate systems became one
    cross = [(a, b) for a, b, ty in want["edges"] if ty == "loop" and (a >> 16) != (b >> 16)]
    assert cross and want["stats"]["optimisations"] >= 1 and len(set(want["coor"])) == 1
