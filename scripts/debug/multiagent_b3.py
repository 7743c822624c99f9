"""debug: is the one-process agents + cloud flow of tests/test_gpu_multiagent.py reproducible run to run, with and without
captured graphs, under the bf16x3 GEMM?"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import test_gpu_multiagent as T
from deeppointmap_amd import knobs
from deeppointmap_amd.system import AgentSystem, CloudSystem

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")


def flow(graphs=True, share=True):
    enc, dec = T._models(dev)
    if not graphs:
        dec.graph_min_hits = 0
    comm = T.LocalComm(); comm.add_member(0)
    for a in (1, 2):
        e, d = (enc, dec) if share else T._models(dev)
        if not graphs:
            d.graph_min_hits = 0
        agent = AgentSystem(T._args(), e, d, system_id=a, comm_module=comm, device=dev)
        agent.start(T._loader(a)); agent.wait()
    uploads = {m["new_scan"]["token"]: m for _, _, c, m in comm.sent if c == "UPLOAD_SCAN"}
    order = sorted(uploads, key=lambda t: (t & 0xffff, t >> 16))
    ce, cd = (enc, dec) if share else T._models(dev)
    if not graphs:
        cd.graph_min_hits = 0
    cloud = CloudSystem(T._args(), ce, cd, comm_module=T.LocalComm(), device=dev)
    for t in order:
        m = uploads[t]
        cloud.step(scan_pack=m["new_scan"], odom_edge=m["odometer_edge"], neighbor_edges=m["neighbor_edges"])
    s = T._cloud_summary(cloud)
    kp = torch.stack([uploads[t]["new_scan"]["key_points"].cpu() for t in order])
    return s, kp


for b3 in (True, False):
    knobs.GEMM_BF16X3 = b3
    ref, kref = flow()
    for name, kw in (("same again", {}), ("no graphs", dict(graphs=False)), ("own models per system", dict(share=False)),
                     ("own models, no graphs", dict(share=False, graphs=False))):
        s, kp = flow(**kw)
        print(f"bf16x3={b3} {name}: descriptors equal {torch.equal(kp, kref)}, poses max diff {np.abs(s['poses'] - ref['poses']).max():.3e}, "
              f"edge max diff {np.abs(s['edge_SE3'] - ref['edge_SE3']).max():.3e}, edges equal {s['edges'] == ref['edges']}")
