"""CPU: the rank-mapped Communicate_Module (deeppointmap_amd/comm.py) over gloo, world_size 3 -- two agents upload scans
to the cloud (member 0) the way SlamSystem.step does (reference system/core.py:411-422), the cloud consumes them the way
CloudSystem's loop does (core.py:520-545) and dismisses the agents."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _scan(agent, i):
    g = torch.Generator().manual_seed(100 * agent + i)
    return dict(new_scan=dict(token=(agent, i), key_points=torch.rand(131, 256, generator=g), full_pcd=torch.rand(3, 1000 + i, generator=g)),
                odometer_edge=dict(src=(agent, i - 1), dst=(agent, i), SE3=torch.rand(4, 4, generator=g),
                                   information_mat=torch.rand(6, 6, generator=g).double()),
                neighbor_edges=[torch.arange(5) + i, "text", 3.5])


def _same(a, b):
    if isinstance(a, torch.Tensor):
        return isinstance(b, torch.Tensor) and a.dtype == b.dtype and torch.equal(a, b)
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return a == b


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deeppointmap_amd.comm import RankCommunicateModule
    comm = RankCommunicateModule(device=torch.device("cpu"))
    for m in range(world):
        comm.add_member(m)
    ok = sorted(comm.get_members()) == list(range(world))
    ok &= comm.fetch_message(rank, block=False) == ("NO_OP", None) and comm.get_queue_length(rank) == 0
    n_scans = 4
    if rank == 0:   # the cloud
        seen = {a: 0 for a in range(1, world)}
        quit_ = set()
        while len(quit_) < world - 1:
            command, message = comm.fetch_message(0, block=True)
            if command == "UPLOAD_SCAN":
                agent, i = message["new_scan"]["token"]
                ok &= i == seen[agent]          # per sender: in the order sent
                ok &= _same(message, _scan(agent, i))
                seen[agent] += 1
            elif command == "AGENT_QUIT":
                quit_.add(message)
            else:
                ok = False
        ok &= all(v == n_scans for v in seen.values())
        for a in range(1, world):
            comm.send_message(caller=0, callee=a, command="QUIT", message=None)
        try:
            comm.send_message(caller=1, callee=0, command="NO_OP", message=None)   # not ours to send
            ok = False
        except ValueError:
            pass
    else:           # an agent
        for i in range(n_scans):
            comm.send_message(caller=rank, callee=0, command="UPLOAD_SCAN", message=_scan(rank, i))
        comm.send_message(caller=rank, callee=rank, command="NO_OP", message="note to self")
        ok &= comm.get_queue_length(rank) >= 1 and comm.fetch_message(rank) == ("NO_OP", "note to self")
        comm.send_message(caller=rank, callee=0, command="AGENT_QUIT", message=rank)
        ok &= comm.fetch_message(rank, block=True) == ("QUIT", None)
    ok &= len(comm.logger) >= 1
    comm.close()
    q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_communicate_module_world3_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)


def _two_channel_worker(rank, world, port, q):
    """separate control and data groups (the shape of the RCCL set-up: control on gloo, tensors on their own group) and a
    payload-free command in the other direction while uploads are in flight"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deeppointmap_amd.comm import RankCommunicateModule
    control, data = dist.new_group(backend="gloo"), dist.new_group(backend="gloo")
    comm = RankCommunicateModule(control_group=control, data_group=data, device=torch.device("cpu"))
    for m in range(world):
        comm.add_member(m)
    ok = True
    if rank == 0:
        comm.send_message(caller=0, callee=1, command="NO_OP", message={"hint": 7})   # cloud -> agent, no tensors
        got = [comm.fetch_message(0, block=True) for _ in range(6)]
        ok &= [g[0] for g in got] == ["UPLOAD_SCAN"] * 6
        ok &= all(_same(g[1], _scan(1, i)) for i, g in enumerate(got))
    else:
        for i in range(6):
            comm.send_message(caller=1, callee=0, command="UPLOAD_SCAN", message=_scan(1, i))
        ok &= comm.fetch_message(1, block=True) == ("NO_OP", {"hint": 7})
    # the RCCL channel's rule (module header): tensor payloads only toward lower member ids, refused BEFORE anything is
    # sent; bare commands go either way.  (RCCL itself cannot run here: the flag is flipped on the gloo-backed module.)
    comm.data_is_nccl = True
    try:
        if rank == 0:
            comm.send_message(caller=0, callee=1, command="UPLOAD_SCAN", message=_scan(0, 0))
            ok = False
    except ValueError:
        pass
    comm.data_is_nccl = False
    try:
        RankCommunicateModule(control_group=None, data_group=None, device=torch.device("cpu")).close()   # a second bus coexists
    except Exception:
        ok = False
    comm.close()
    q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_communicate_module_separate_data_group_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_two_channel_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res)
