"""debug: where do the 3-process and the 1-process runs of tests/test_gpu_multiagent.py part?"""
import hashlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_multiagent as M


def h(t):
    if t is None:
        return "-"
    return hashlib.md5(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:8]


def hook_reg(dec, rec, cloud=None):
    f0 = dec.registration_forward

    def f(src, dst, **kw):
        if cloud is not None:
            rec.append(("poses", [(t, h(p)) for t, p in cloud.backend.poses.items()], [h(dst[128:, i * 128:(i + 1) * 128]) for i in range(dst.shape[1] // 128)],
                        [h(dst[:128, i * 128:(i + 1) * 128]) for i in range(dst.shape[1] // 128)]))
        R, T, conf, rmse = f0(src, dst, **kw)
        rec.append(("regio", tuple(src.shape), h(src), h(dst), h(R), h(T), h(conf), rmse))
        return R, T, conf, rmse
    dec.registration_forward = f


def hook(cloud, rec):
    from deeppointmap_amd.posegraph_optim import optimize_pose_graph
    import numpy as np

    def opt(nodes, es, base):
        hn = hashlib.md5(b"".join(np.ascontiguousarray(v).tobytes() for v in nodes.values())).hexdigest()[:8]
        he = hashlib.md5(b"".join(np.ascontiguousarray(e[2]).tobytes() + np.ascontiguousarray(e[3]).tobytes() for e in es)).hexdigest()[:8]
        out, _ = optimize_pose_graph(nodes, es, base_token=base)
        ho = hashlib.md5(b"".join(np.ascontiguousarray(v).tobytes() for v in out.values())).hexdigest()[:8]
        rec.append((hn, he, ho))
        return out
    cloud.backend.optimiser = opt


def digest(m):
    s, o = m["new_scan"], m["odometer_edge"]
    d = dict(tok=s["token"], kp=h(s["key_points"]), pcd=h(s["full_pcd"]), pose=h(s["SE3_pred"]))
    if o is not None:
        d.update(o_SE3=h(o["SE3"]), o_info=h(o["information"]), o_rmse=o["rmse"], o_conf=o["confidence"])
    d["nb"] = [(e["src"], e["dst"], h(e["SE3"]), h(e["information"])) for e in m["neighbor_edges"]]
    return d


def run(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_grad_enabled(False)
    from deeppointmap_amd.comm import RankCommunicateModule
    from deeppointmap_amd.system import AgentSystem, CloudSystem
    dev = torch.device("cuda:0")
    enc, dec = M._models(dev)
    comm = RankCommunicateModule(device=dev)
    for m in range(world):
        comm.add_member(m)
    if rank == 0:
        cloud = CloudSystem(M._args(), enc, dec, comm_module=comm, device=dev, keep_log=True)
        digs = []
        rec = []
        hook(cloud, rec)
        hook_reg(dec, rec, cloud)
        step0 = cloud.step

        def step(scan_pack, odom_edge, neighbor_edges):
            digs.append(digest(dict(new_scan=scan_pack, odometer_edge=odom_edge, neighbor_edges=neighbor_edges)))
            r = step0(scan_pack=scan_pack, odom_edge=odom_edge, neighbor_edges=neighbor_edges)
            digs[-1]["after"] = [(t, h(p)) for t, p in cloud.backend.poses.items()]
            return r
        cloud.step = step
        cloud.start(expected_agents=world - 1)
        cloud.wait()
        for a in range(1, world):
            comm.send_message(caller=0, callee=a, command="QUIT", message=None)
        log = [(c["kind"], h(c.get("SE3")), h(c.get("G")), h(c.get("prob"))) for c in cloud.backend.log]
        q.put((cloud.arrivals, digs, log, rec))
    else:
        agent = AgentSystem(M._args(), enc, dec, system_id=rank, comm_module=comm, device=dev)
        agent.start(M._loader(rank))
        agent.wait()
        comm.send_message(caller=rank, callee=0, command="AGENT_QUIT", message=rank)
        comm.fetch_message(rank, block=True)
    torch.cuda.synchronize()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    from deeppointmap_amd.system import AgentSystem, CloudSystem
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=run, args=(r, 3, 29871, q)) for r in range(3)]
    [p.start() for p in procs]
    order, digs, log, rec3 = q.get(timeout=240)
    [p.join() for p in procs]
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    enc, dec = M._models(dev)
    comm = M.LocalComm()
    comm.add_member(0)
    for a in (1, 2):
        agent = AgentSystem(M._args(), enc, dec, system_id=a, comm_module=comm, device=dev)
        agent.start(M._loader(a))
        agent.wait()
    uploads = {m["new_scan"]["token"]: m for _, _, c, m in comm.sent if c == "UPLOAD_SCAN"}
    cloud = CloudSystem(M._args(), enc, dec, comm_module=M.LocalComm(), device=dev, keep_log=True)
    rec1 = []
    hook(cloud, rec1)
    hook_reg(dec, rec1, cloud)
    for i, t in enumerate(order):
        m = uploads[t]
        d = digest(m)
        cloud.step(scan_pack=m["new_scan"], odom_edge=m["odometer_edge"], neighbor_edges=m["neighbor_edges"])
        d["after"] = [(t_, h(p)) for t_, p in cloud.backend.poses.items()]
        if d != digs[i]:
            print("arrival", i, "token", t, "differs")
            for k in d:
                if d[k] != digs[i][k]:
                    print("   ", k, "\n      one:", d[k], "\n      3p :", digs[i][k])
        else:
            print("arrival", i, "token", t, "equal")
    log1 = [(c["kind"], h(c.get("SE3")), h(c.get("G")), h(c.get("prob"))) for c in cloud.backend.log]
    for i, (a, b) in enumerate(zip(log1, log)):
        print(i, "==" if a == b else "!=", a, b)
    for a, b in zip(rec1, rec3):
        print("optim (nodes, edges, out):", a, b)
