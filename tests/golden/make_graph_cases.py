#!/usr/bin/env python3
"""Pose-graph queries of the REFERENCE on random graphs -> tests/golden/graph_cases.json: PoseGraph.graph_search
(pose_graph.py:513-542), shortest_path_length (:544-563), repair_coor_sys (:844-864) and the scan selection of
global_map_query_graph (:491-496).  Runs only in the build container (imports /root/reference, read-only; colorlog,
easydict, readerwriterlock and open3d are stubbed as in make_trace.py -- none of the stubbed code runs)."""
import json
import logging
import math
import os
import random
import sys
import tempfile
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules["colorlog"] = logging
sys.modules.setdefault("open3d", types.ModuleType("open3d"))
ed = types.ModuleType("easydict")
ed.EasyDict = dict
sys.modules["easydict"] = ed
rw, rwl = types.ModuleType("readerwriterlock"), types.ModuleType("readerwriterlock.rwlock")


class _L:
    def acquire(self, blocking=True):
        return True

    def release(self):
        pass


class RWLockFair:
    def gen_rlock(self):
        return _L()

    def gen_wlock(self):
        return _L()


rwl.RWLockFair = RWLockFair
rw.rwlock = rwl
sys.modules["readerwriterlock"], sys.modules["readerwriterlock.rwlock"] = rw, rwl
sys.path.insert(0, "/root/reference")
from system.modules.pose_graph import PoseGraph, PoseGraph_Edge, ScanPack  # noqa: E402
from system.modules.recoder import ResultLogger  # noqa: E402


def main():
    rng = random.Random(7)
    cases = []
    for c in range(12):
        n_agents = 1 if c < 6 else 3
        pg = PoseGraph(args=None, agent_id=0, device="cpu")
        scans, edges = [], []
        per = rng.randint(8, 30)
        for a in range(n_agents):
            pos = torch.tensor([rng.uniform(-20, 20), rng.uniform(-20, 20), 0.0])
            prev_kf = None
            for s in range(per):
                pos = pos + torch.tensor([rng.uniform(0.5, 6.0), rng.uniform(-2, 2), rng.uniform(-0.2, 0.2)])
                SE3 = torch.eye(4)
                SE3[:3, 3] = pos
                if c % 2 == 0:      # headings too (the text writers print rotations)
                    yaw = rng.uniform(-3.0, 3.0)
                    SE3[0, 0], SE3[0, 1], SE3[1, 0], SE3[1, 1] = math.cos(yaw), -math.sin(yaw), math.sin(yaw), math.cos(yaw)
                sp = ScanPack(timestamp=s * 0.1, agent_id=a, timestep=s, key_points=torch.zeros(4, 2), SE3_pred=SE3, coor_sys=a)
                kf = prev_kf is None or rng.random() < 0.7
                if not kf:
                    sp = sp.nonkeyframe()
                pg.add_vertex(sp)
                scans.append(dict(token=sp.token, type=sp.type, xyz=pos.tolist(), coor=a, SE3=SE3.tolist()))
                if prev_kf is not None:
                    ty = "odom" if kf else "locz"
                    E = torch.eye(4)
                    E[:3, 3] = torch.tensor([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-0.1, 0.1)])
                    info = torch.diag(torch.tensor([rng.uniform(1, 50) for _ in range(6)]))
                    pg.add_edge(PoseGraph_Edge(prev_kf, sp.token, E, info, type=ty))
                    edges.append((prev_kf, sp.token, ty, E.tolist(), info.tolist()))
                if kf:
                    prev_kf = sp.token
        kfs = [s["token"] for s in scans if s["type"] == "full"]
        for _ in range(rng.randint(0, 6)):                      # loop edges, within and between agents
            a, b = rng.sample(kfs, 2)
            if not pg.has_edge(a, b) and not pg.has_edge(b, a):
                pg.add_edge(PoseGraph_Edge(a, b, torch.eye(4), torch.eye(6), type="loop"))
                edges.append((a, b, "loop", torch.eye(4).tolist(), torch.eye(6).tolist()))
        q = []
        for _ in range(25):
            t = rng.choice(kfs)
            level, max_k = rng.choice([2, 3, 5, 30]), rng.choice([None, 16, 4])
            kinds = rng.choice([["odom", "loop"], "all", ["odom"]])
            q.append(dict(fn="graph_search", token=t, level=level, max_k=max_k, kinds=kinds,
                          out=[s.token for s in pg.graph_search(t, level, coor_sys=0, edge_type=kinds, max_k=max_k)]))
            u = rng.choice(kfs)
            inf = rng.choice([5, 50, 5000])
            q.append(dict(fn="shortest", src=t, dst=u, kinds=kinds, inf=inf,
                          out=pg.shortest_path_length(t, u, edge_type=kinds, infinity_length=inf)))
            md = rng.choice([None, 20, 8])
            sel = [s for s in pg.graph_search(token=t, neighbor_level=5, coor_sys=0, edge_type=["odom", "loop"]) if s.type != "non-keyframe"]
            if md is not None:
                ct = pg.get_scanpack(t).SE3_pred[:3, 3:]
                sel = [s for s in sel if torch.norm(s.SE3_pred[:3, 3:] - ct, p=2, dim=0).item() < md]
            q.append(dict(fn="map_tokens", token=t, max_dist=md, out=[s.token for s in sel]))
        files = {}
        if c < 4:       # the reference's text writers on this graph (recoder.py:76-97 trajectory, pose_graph.py:821-842 g2o)
            with tempfile.TemporaryDirectory() as d:
                rl = ResultLogger(args=None, system_info=None, posegraph_map=pg, log_dir=d)
                rl.save_trajectory("traj")
                rl.save_posegraph("graph")
                files = {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d))}
        pg.repair_coor_sys()
        cases.append(dict(scans=scans, edges=edges, queries=q, files=files, coor_after={str(s.token): s.coor_sys for s in pg.get_all_scans()}))
    with open(os.path.join(HERE, "graph_cases.json"), "w") as f:
        json.dump(cases, f)
    print(len(cases), "graphs,", sum(len(c["queries"]) for c in cases), "queries")


if __name__ == "__main__":
    main()
