"""The sequential part of the SLAM front/back end that stays on rank 0 after the frame-sharded encode: what
SlamSystem.step does with a new scan once its descriptors and its speculative odometry edge exist (reference
system/core.py:382-407):

  * MappingThread.valid_check (mapping.py:52-83): an edge below `edge_confidence_drop` or above `edge_rmse_drop` drops its
    scan; after `max_continuous_drop_scan` drops in a row the best of the bag is recovered;
  * MappingThread.keyframe_check (mapping.py:85-134): the new pose = key-frame pose @ edge; the scan is a key-frame unless a
    key-frame of the graph neighbourhood (PoseGraph.graph_search: breadth first over odometry / loop edges, five levels, at
    most 16 scans; pose_graph.py:513-542) lies within `key_frame_distance` (fixed, or 'auto': the running ratio of
    mapping.py:86-92);
  * MappingThread.scan_to_map_adjustment (mapping.py:136-170) for every key-frame: the tile of the same graph neighbourhood
    (device resident: maptile.MapTileStore), centred on the previous key-frame, minus the new scan's own columns, registered
    against the new scan; accepted under mapping.py:193-201;
  * the pose-graph optimisation that follows loop closures (loop_closure.py:294-307).  Loop DETECTION is not rebuilt here:
    `add_loop_edge` takes edges from whoever ran it (tests: the reference's recorded ones); bench.py, which only needs the
    cost of the optimiser, closes a loop over the last `optimize_every` key-frames.

It consumes what shard.gather_step_results delivers -- descriptors (n,131,S) and edge rows (n,EDGE_FLOATS) in sequence
order.  One difference of the sharded path is visible here: the ranks register every frame against its PREDECESSOR
(speculatively, before anybody knows which frames become key-frames), the reference against the last KEY-frame
(odometry.py:82-97).  While the predecessor is the key-frame the two are the same edge; otherwise the edge to the key-frame is
the product of the consecutive edges since then (`exact_odometry=True` re-registers against the key-frame on this rank
instead, as the reference would: one more 256 x 256 registration per non-key-frame).

This is the Amdahl term of the multi-GPU path: bench.py reports its cost per gathered step for N > 1.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .maptile import MapTileStore
from .posegraph_optim import optimize_pose_graph
from .registration import PoseTool, simvec_to_num

ACPT, DROP, DIST = "acpt", "drop", "dist"  # EXIT_CODE of the reference (system/modules/utils.py)


def default_slam_args() -> dict:
    """configs/infer/DeepPointMap_B_Main_SemanticKITTI.yaml:63-80 (mapping part)"""
    return dict(edge_confidence_drop=0.60, edge_rmse_drop=0.50, max_continuous_drop_scan=5,
                continuous_drop_scan_strategy="recover", key_frame_distance="auto", key_frame_distance_0=10.0,
                enable_s2m_adjust=True, registration_sample_mapping=0.5, registration_sample_odometer=0.5)


class Rank0Consumer:
    def __init__(self, decoder, device, slam_args: Optional[dict] = None, optimize_every: int = 0,
                 exact_odometry: bool = False):
        self.decoder, self.device = decoder, torch.device(device)
        self.args = dict(default_slam_args(), **(slam_args or {}))
        self.optimize_every, self.exact_odometry = optimize_every, exact_odometry
        self.store = MapTileStore(self.device)
        self.type: Dict[int, str] = {}                 # token -> 'full' (key-frame) | 'non-keyframe'
        self.poses: Dict[int, torch.Tensor] = {}       # token -> SE3_pred (4,4) CPU
        self.edges: Dict[Tuple[int, int], dict] = {}   # (src, dst) -> SE3, type, information, confidence, rmse
        self.adj: Dict[int, List[Tuple[int, dict]]] = {}   # token -> (neighbour, edge) in edge insertion order
        self.desc: Dict[int, torch.Tensor] = {}        # key-frames and the last frame: descriptors on the device
        self.last_known_keyframe: Optional[int] = None
        self.last_known_anyframe: Optional[int] = None
        self.n_frames = 0
        self.codes: List[str] = []
        self.tiles: List[List[int]] = []               # token order of every scan-to-map tile (tests)
        self.drop_bag: List[tuple] = []
        self.since_kf = torch.eye(4)                   # product of the consecutive edges since the last key-frame
        self.chain_ok = True
        if self.args["key_frame_distance"] == "auto":
            self.dist_auto, self.dist_ratio = True, 1.0
            self.kf_dist0 = float(self.args.get("key_frame_distance_0", 3.0))
            self.cur_kf_dist = self.kf_dist0 * self.dist_ratio
        else:
            self.dist_auto, self.kf_dist0 = False, float(self.args["key_frame_distance"])
            self.cur_kf_dist = self.kf_dist0
        self.stats = dict(s2m=0, optimisations=0, keyframes=0, dropped=0, re_registrations=0)

    # -- the pose graph, as much of it as the gating needs ---------------------------------------------------------
    @property
    def keyframes(self) -> List[int]:
        return [t for t, ty in self.type.items() if ty != "non-keyframe"]

    def _add_edge(self, src: int, dst: int, edge: dict) -> None:
        self.edges[(src, dst)] = edge
        self.adj.setdefault(src, []).append((dst, edge))
        self.adj.setdefault(dst, []).append((src, edge))

    def _neighbors(self, tok: int, kinds) -> List[int]:
        """PoseGraph.get_neighbor_tokens filtered by edge type: the other ends of the scan's edges, in the order the edges
        entered the graph (the reference walks its edge dict; an adjacency list per scan gives the same order)"""
        return [n for n, e in self.adj.get(tok, ()) if e["type"] in kinds]

    def graph_search(self, tok: int, level: int = 5, kinds=("odom", "loop"), max_k: int = 16) -> List[int]:
        """PoseGraph.graph_search (pose_graph.py:513-542)"""
        found: Dict[int, None] = {}
        bfs = [(level, tok)]
        while bfs and len(found) < max_k:
            rem, t = bfs.pop(0)
            if t in found:
                continue
            found[t] = None
            if rem <= 0:
                continue
            bfs += [(rem - 1, n) for n in self._neighbors(t, kinds)]
        return list(found)

    def add_loop_edge(self, src: int, dst: int, SE3: torch.Tensor, information=None, confidence: float = 1.0, rmse: float = 0.0):
        self._add_edge(src, dst, dict(SE3=SE3.clone(), type="loop", information=information, confidence=confidence, rmse=rmse))

    def optimise(self, tokens: Optional[List[int]] = None) -> None:
        """PoseGraph.optim on the key-frames (all, or `tokens`) and the non-'locz' edges between them."""
        toks = [t for t in (tokens if tokens is not None else self.keyframes) if self.type.get(t) != "non-keyframe"]
        if len(toks) < 2:
            return
        nodes = {t: self.poses[t].double().numpy() for t in toks}
        es = [(a, b, e["SE3"].double().numpy(),
               np.asarray(e["information"] if e["information"] is not None else np.eye(6), dtype=np.float64))
              for (a, b), e in self.edges.items() if e["type"] != "locz" and a in nodes and b in nodes]
        refined, _ = optimize_pose_graph(nodes, es, base_token=min(toks))
        for t, P in refined.items():
            self.poses[t] = torch.from_numpy(np.asarray(P, dtype=np.float32))
        self.stats["optimisations"] += 1

    # -- MappingThread ---------------------------------------------------------------------------------------------------
    def _valid_check(self, tok: int, edge: dict):
        a = self.args
        if edge["confidence"] < a["edge_confidence_drop"] or edge["rmse"] > a["edge_rmse_drop"]:
            self.drop_bag.append((tok, edge))
            if len(self.drop_bag) >= a["max_continuous_drop_scan"]:
                if a["continuous_drop_scan_strategy"] != "recover":
                    raise NotImplementedError("continuous_drop_scan_strategy 'break' (mapping.py:65-74) is not mapped")
                tok, edge = min(self.drop_bag, key=lambda x: x[1]["rmse"])
                self.drop_bag.clear()
                return ACPT, tok, edge
            return DROP, tok, edge
        self.drop_bag.clear()
        return ACPT, tok, edge

    def _keyframe_check(self, tok: int, edge: dict) -> str:
        a = self.args
        if self.dist_auto:
            rmse_ratio = min(edge["rmse"] / a["edge_rmse_drop"], 1.0)
            self.dist_ratio = max(min(0.90 * self.dist_ratio + 0.10 * ((1 - rmse_ratio) ** 2) * 2.0, 2.0), 0.0)
            self.cur_kf_dist = max(self.kf_dist0 * self.dist_ratio, 1.0)
        old = edge["src"]
        self.poses[tok] = self.poses[old] @ edge["SE3"]
        self.last_known_keyframe = old
        if self.cur_kf_dist >= 0:
            near = [t for t in self.graph_search(old) if self.type[t] != "non-keyframe"]
            d = torch.stack([self.poses[t][:3, 3] for t in near]) - self.poses[tok][:3, 3].unsqueeze(0)
            if float(torch.norm(d, p=2, dim=1).min()) < self.cur_kf_dist:
                return DIST
        return ACPT

    def _scan_to_map(self, tok: int, edge: dict) -> dict:
        if not self.args["enable_s2m_adjust"]:
            return edge
        old = edge["src"]
        toks = [t for t in self.graph_search(old) if self.type[t] != "non-keyframe"]
        self.tiles.append(list(toks))
        tile, owner = self.store.tile(toks, [self.poses[t] for t in toks], self.poses[old])
        src = tile[:, (owner != tok).to(self.device)]                    # "drop same descriptors from map" (mapping.py:146)
        R, T, conf, rmse = self.decoder.registration_forward(src, self.desc[tok], num_sample=self.args["registration_sample_mapping"])
        self.stats["s2m"] += 1
        return dict(src=old, dst=tok, SE3=PoseTool.SE3(R.cpu(), T.cpu()).inverse(), type="odom", information=edge["information"],
                    confidence=simvec_to_num(conf), rmse=rmse)

    @torch.no_grad()
    def push(self, tok: int, desc: torch.Tensor, row: torch.Tensor) -> str:
        """One scan in sequence order: desc (131,S) on the device, row = its gathered edge row (host): the registration of
        its predecessor (source) against it.  Returns the reference's exit code ('acpt' / 'drop' / 'dist')."""
        self.n_frames += 1
        if self.last_known_keyframe is None:            # first scan of the graph (core.py:383-388)
            self.poses[tok], self.type[tok], self.desc[tok] = torch.eye(4), "full", desc
            self.store.put(tok, desc)
            self.last_known_anyframe = self.last_known_keyframe = tok
            self.stats["keyframes"] += 1
            self.codes.append(ACPT)
            return ACPT
        kf = self.last_known_keyframe
        step = PoseTool.SE3(row[0:9].view(3, 3), row[9:12].view(3, 1)).inverse()   # edge.SE3 = registration^-1 (odometry.py:119)
        conf30, rmse = float(row[16]), float(row[12])
        info = row[ops.RES_HDR:ops.RES_HDR + 36].view(6, 6).clone()
        if self.chain_ok and not self.exact_odometry:
            rel = self.since_kf @ step                  # key-frame -> predecessor -> this scan
        else:                                           # the reference's own edge: this scan against the key-frame
            R, T, conf, rmse = self.decoder.registration_forward(self.desc[kf], desc, num_sample=self.args["registration_sample_odometer"])
            rel, conf30 = PoseTool.SE3(R.cpu(), T.cpu()).inverse(), simvec_to_num(conf)
            self.stats["re_registrations"] += 1
        edge = dict(src=kf, dst=tok, SE3=rel, type="odom", information=info, confidence=conf30, rmse=rmse)
        self.desc[tok] = desc
        code, tok, edge = self._valid_check(tok, edge)
        if code != ACPT:
            self.stats["dropped"] += 1
            self.chain_ok = False                       # the next scan's consecutive edge hangs on a dropped scan
            self.codes.append(code)
            return code
        self.last_known_keyframe = edge["src"]
        code = self._keyframe_check(tok, edge)
        if code != ACPT:
            self.type[tok] = "non-keyframe"
            self.last_known_anyframe = tok
            self._add_edge(edge["src"], tok, dict(edge, type="locz"))
            self.since_kf, self.chain_ok = edge["SE3"].clone(), True
            self.codes.append(code)
            return code
        self.type[tok] = "full"
        self.store.put(tok, self.desc[tok])
        self.last_known_anyframe = self.last_known_keyframe = tok
        self._add_edge(edge["src"], tok, dict(edge))
        self.stats["keyframes"] += 1
        new = self._scan_to_map(tok, edge)
        if new["rmse"] <= self.args["edge_rmse_drop"] or new["rmse"] <= edge["rmse"]:     # mapping.py:193-201
            self.poses[tok] = self.poses[new["src"]] @ new["SE3"]
            self.edges[(edge["src"], tok)].update(SE3=new["SE3"], confidence=new["confidence"], rmse=new["rmse"])
        for t in [t for t in self.desc if t != tok and self.type.get(t) == "non-keyframe"]:
            del self.desc[t]                            # non-key-frames are never registered against again
        self.since_kf, self.chain_ok = torch.eye(4), True
        if self.optimize_every and self.stats["keyframes"] % self.optimize_every == 0:
            window = self.keyframes[-self.optimize_every:]
            a = window[0]
            if (a, tok) not in self.edges and (tok, a) not in self.edges:
                self.add_loop_edge(a, tok, torch.linalg.inv(self.poses[a]) @ self.poses[tok], torch.eye(6) * 10.0)
            self.optimise(window)
        self.codes.append(ACPT)
        return ACPT

    @torch.no_grad()
    def consume(self, desc: torch.Tensor, table: torch.Tensor) -> float:
        """One gathered step: desc (n,131,S), table (n,EDGE_FLOATS) on the device, frames in sequence order.
        Returns the wall time in ms (device work included: the caller's results are only final after it)."""
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        head = table[:, :ops.RES_HDR + 36].cpu()       # the one download: 56 floats per frame
        base = self.n_frames
        for g in range(desc.shape[0]):
            self.push(base + g, desc[g], head[g])
        torch.cuda.synchronize(self.device)
        return (time.perf_counter() - t0) * 1e3
