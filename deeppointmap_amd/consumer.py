"""The sequential part of the SLAM front / back end that stays on rank 0 after the frame-sharded encode: everything
SlamSystem.step does with a new scan once its descriptors exist (reference system/core.py:382-407), with every
registration, loop-detection batch, map tile and information matrix on the device path:

  * OdometryThread (odometry.py:76-131): the key-frame of the graph neighbourhood nearest to the last known pose is the
    registration partner; edge.SE3 = registration^-1, information matrix from the two full clouds;
  * MappingThread.valid_check (mapping.py:52-83): an edge below `edge_confidence_drop` or above `edge_rmse_drop` drops its
    scan; the `max_continuous_drop_scan`-th drop in a row is accepted instead ('recover': the reference picks the best of
    the bag into LOCAL names and then carries on with the scan at hand -- so do we);
  * MappingThread.keyframe_check (mapping.py:85-134): the new pose = key-frame pose @ edge; the scan is a key-frame unless a
    key-frame of the graph neighbourhood (PoseGraph.graph_search: breadth first over odometry / loop edges, five levels, at
    most 16 scans; pose_graph.py:513-542) lies within `key_frame_distance` (fixed, or 'auto': the running ratio of
    mapping.py:86-92);
  * MappingThread.scan_to_map_adjustment (mapping.py:136-170) for every key-frame: the tile of the same graph neighbourhood
    within 20 m (pose_graph.py:471-511; device resident: maptile.MapTileStore), centred on the previous key-frame, minus
    the new scan's own columns, registered against the new scan; accepted under mapping.py:193-201;
  * LoopThread.process (loop_closure.py:56-307) for every key-frame: candidates = key-frames outside the trusted zones and
    inside the search radius, one `loop_detection_forward` batch, the top-k above the probability threshold; for each, the
    two neighbourhood tiles with the shared scans dealt to the nearer side, one map-to-map registration, the information
    matrix of the two scans; verification by confidence and by the disagreement with the graph path; accepted edges enter
    the graph and the optimiser runs under the reference's gap / flag rules;
  * PoseGraph.optim (pose_graph.py:565-658): every key-frame, every non-'locz' edge, reference node = the oldest scan; the
    non-key-frames are re-hung on their key-frames afterwards.  The solver is posegraph_optim (open3d's algorithm restated:
    parity unpinned); `optimiser=` replaces it (tests record the call and return nothing).

Two ways in.  `push(tok, desc, row, pcd)` consumes what shard.gather_step_results delivers -- descriptors and edge rows in
sequence order.  One difference of the sharded path is visible there: the ranks register every frame against its
PREDECESSOR (speculatively, before anybody knows which frames become key-frames), the reference against a key-frame.  While
the predecessor is that key-frame the two are the same edge; otherwise the edge to the key-frame is the product of the
consecutive edges since then (`exact_odometry=True`, or `row=None`, registers against the key-frame on this rank instead, as
the reference would: one more 256 x 256 registration).  `step(desc, pcd)` is the reference's own flow for one agent: no
rows, tokens counted here.

This is the Amdahl term of the multi-GPU path: bench.py reports its cost per gathered step for N > 1.
"""
from __future__ import annotations

import time
from collections import OrderedDict
from math import sqrt
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .maptile import MapTileStore
from .posegraph_optim import optimize_pose_graph
from .registration import PoseTool, simvec_to_num

ACPT, DROP, DIST = "acpt", "drop", "dist"  # EXIT_CODE of the reference (system/modules/utils.py)
TRANS_STD, ROT_STD = 0.4, 0.5              # LoopThread.TRANS_STD / ROT_STD (loop_closure.py:16-17)


def se3_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a @ b for 4x4 poses, every entry summed in double in index order and rounded once.  The reference multiplies float32
    tensors through the BLAS of the day, whose last bit depends on the thread and the process it runs in (seen on the test
    box: the cloud thread of a spawned rank against the main thread of another process, same inputs); a trajectory must not."""
    A, B = a.double().tolist(), b.double().tolist()
    return torch.tensor([[A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j] + A[i][3] * B[3][j] for j in range(4)]
                         for i in range(4)], dtype=torch.float32)


def se3_inv(a: torch.Tensor) -> torch.Tensor:
    """inverse of a rigid 4x4 pose: [R^T, -R^T t], in double, rounded once (the reference calls the general LU inverse:
    the same matrix to rounding, and as order-dependent as its products)"""
    A = a.double().tolist() if isinstance(a, torch.Tensor) else a      # (nested lists: rows of float32 values as floats)
    Rt = [[A[j][i] for j in range(3)] for i in range(3)]
    t = [-(Rt[i][0] * A[0][3] + Rt[i][1] * A[1][3] + Rt[i][2] * A[2][3]) for i in range(3)]
    return torch.tensor([Rt[0] + [t[0]], Rt[1] + [t[1]], Rt[2] + [t[2]], [0.0, 0.0, 0.0, 1.0]], dtype=torch.float32)


def default_slam_args() -> dict:
    """configs/infer/DeepPointMap_B_Main_SemanticKITTI.yaml:63-97"""
    return dict(coor_scale=60, odometer_candidates_num=1, registration_sample_odometer=0.5,
                edge_confidence_drop=0.60, edge_rmse_drop=0.50, max_continuous_drop_scan=5,
                continuous_drop_scan_strategy="recover", key_frame_distance="auto", key_frame_distance_0=10.0,
                enable_s2m_adjust=True, registration_sample_mapping=0.5,
                enable_loop_closure=True, loop_detection_gap=0, loop_detection_transaction_gap=10.0,
                loop_detection_trust_range=3, loop_detection_gnss_distance=-1, loop_detection_pred_distance=100.0,
                loop_detection_rotation_min=30.0, loop_detection_translation_min=10.0,
                loop_detection_prob_acpt_threshold=0.7, loop_detection_candidates_num=1, registration_sample_loop=0.5,
                loop_detection_confidence_acpt_threshold=0.6, enable_global_optimization=True, global_optimization_gap=0)


class ScanCloudStore:
    """token -> full cloud (3,N) in metres, for the information matrices of the edges (utils.py:60-113).  The reference keeps every
    ScanPack.full_pcd on the host for the whole run (pose_graph.py:43-69); here the clouds the odometry / mapping / loop steps
    are about to pair stay on the DEVICE, and their total is bounded: beyond `max_device_bytes` the least recently used clouds
    move to host memory and come back (same bits) when an edge asks for them -- a loop closure against a key-frame of an hour
    ago pays one upload, a long drive does not grow the GPU footprint without bound."""

    def __init__(self, device, max_device_bytes: int = 4 << 30):
        self.device, self.max_device_bytes = torch.device(device), int(max_device_bytes)
        self._dev: "OrderedDict[int, torch.Tensor]" = OrderedDict()   # least recently used first
        self._host: Dict[int, torch.Tensor] = {}
        self._bytes = 0
        self.stats = dict(evicted=0, restored=0)

    def __contains__(self, tok) -> bool:
        return tok in self._dev or tok in self._host

    def __len__(self) -> int:
        return len(self._dev) + len(self._host)

    def device_bytes(self) -> int:
        return self._bytes

    def __setitem__(self, tok, pcd: torch.Tensor) -> None:
        self.pop(tok, None)
        pcd = pcd if pcd.device == self.device else pcd.to(self.device)
        self._dev[tok] = pcd
        self._bytes += pcd.numel() * pcd.element_size()
        self._shrink(keep=tok)

    def _shrink(self, keep) -> None:
        while self._bytes > self.max_device_bytes and len(self._dev) > 1:
            tok = next(iter(self._dev))
            if tok == keep:
                self._dev.move_to_end(tok)
                continue
            t = self._dev.pop(tok)
            self._bytes -= t.numel() * t.element_size()
            self._host[tok] = t.cpu()           # synchronous copy: the device block is free for reuse when it returns
            self.stats["evicted"] += 1

    def __getitem__(self, tok) -> torch.Tensor:
        if tok in self._dev:
            self._dev.move_to_end(tok)
            return self._dev[tok]
        t = self._host.pop(tok).to(self.device)
        self.stats["restored"] += 1
        self._dev[tok] = t
        self._bytes += t.numel() * t.element_size()
        self._shrink(keep=tok)
        return t

    def get(self, tok, default=None):
        return self[tok] if tok in self else default

    def pop(self, tok, default=None):
        if tok in self._dev:
            t = self._dev.pop(tok)
            self._bytes -= t.numel() * t.element_size()
            return t
        return self._host.pop(tok, default)


class Rank0Consumer:
    def __init__(self, decoder, device, slam_args: Optional[dict] = None, optimize_every: int = 0,
                 exact_odometry: bool = False, agent_id: int = 0, loop_targets: str = "self",
                 optimiser: Optional[Callable] = None, keep_log: bool = False):
        """optimize_every > 0: bench.py's stand-in for a loop closure (it only needs the optimiser's cost): every that many
        key-frames an edge closes the window and the window is optimised; the reference's own loop closure is then off
        unless `slam_args` turns it on."""
        self.decoder, self.device = decoder, torch.device(device)
        self.args = dict(default_slam_args(), **(slam_args or {}))
        if optimize_every and "enable_loop_closure" not in (slam_args or {}):
            self.args["enable_loop_closure"] = False
        self.optimize_every, self.exact_odometry = optimize_every, exact_odometry
        self.agent_id, self.loop_targets, self.optimiser, self.keep_log = agent_id, loop_targets, optimiser, keep_log
        self.store: Optional[MapTileStore] = None      # key-frame descriptors in HBM (sized by the first one)
        self.type: Dict[int, str] = {}                 # token -> 'full' (key-frame) | 'non-keyframe', in insertion order
        self.poses: Dict[int, torch.Tensor] = {}       # token -> SE3_pred (4,4) CPU
        self.edges: Dict[Tuple[int, int], dict] = {}   # (src, dst) -> SE3, type, information, confidence, rmse
        self.adj: Dict[int, List[Tuple[int, dict]]] = {}   # token -> (neighbour, edge) in edge insertion order
        self._searches: Dict[tuple, List[int]] = {}    # graph_search results since the last new edge of their kinds
        self._near: Dict[int, tuple] = {}              # key-frame neighbourhood of a scan + its positions (see _neighbourhood)
        self.desc: Dict[int, torch.Tensor] = {}        # key-frames and the scan at hand: descriptors on the device
        # full clouds (3,N) in metres, when the caller has them: on the device up to a budget, the rest on the host
        self.pcd = ScanCloudStore(self.device, int(self.args.get("scan_cloud_device_bytes", 4 << 30)))
        self.coor: Dict[int, int] = {}                 # token -> coordinate system (multi-agent: one per agent until loops merge them)
        self.coor_sys = agent_id                       # SlamSystem.coor_sys (core.py:44)
        self.last_known_keyframe: Optional[int] = None
        self.last_known_anyframe: Optional[int] = None
        self.key_frame_num = 0
        self.n_frames = 0
        self.codes: List[str] = []
        self.tiles: List[List[int]] = []               # keep_log: token order of every scan-to-map tile (tests)
        self.log: List[dict] = []                      # keep_log: every device call, in order (tests hold it to the reference's)
        self.drop_bag: List[tuple] = []
        self.since_kf = torch.eye(4)                   # product of the consecutive edges since ...
        self.chain_base: Optional[int] = None          # ... this key-frame
        self.chain_ok = True
        if self.args["key_frame_distance"] == "auto":
            self.dist_auto, self.dist_ratio = True, 1.0
            self.kf_dist0 = float(self.args.get("key_frame_distance_0", 3.0))
            self.cur_kf_dist = self.kf_dist0 * self.dist_ratio
        else:
            self.dist_auto, self.kf_dist0 = False, float(self.args["key_frame_distance"])
            self.cur_kf_dist = self.kf_dist0
        a = self.args
        self.last_loop_pose_num = -a["loop_detection_gap"] - 1
        self.last_optim_pose_num = -a["global_optimization_gap"] - 1
        self.last_loop_token = -1
        self.required_optim = False
        self.stats = dict(s2m=0, optimisations=0, keyframes=0, dropped=0, re_registrations=0, loop_batches=0,
                          loop_registrations=0, loop_edges=0)

    # -- the pose graph ------------------------------------------------------------------------------------------------
    @property
    def keyframes(self) -> List[int]:
        return [t for t, ty in self.type.items() if ty != "non-keyframe"]

    def _add_vertex(self, tok: int, kind: str, coor: Optional[int] = None) -> None:
        assert tok not in self.type, f"Scan {tok} already in posegraph map"
        self.type[tok] = kind
        self.coor[tok] = self.coor_sys if coor is None else coor
        if kind == "full":
            self.key_frame_num += 1
            self.stats["keyframes"] += 1

    def _add_edge(self, src: int, dst: int, edge: dict) -> None:
        if (src, dst) in self.edges or (dst, src) in self.edges:
            raise RuntimeError(f"Received an edge that already exists ({src} - {dst})")   # pose_graph.py:198-205
        self.edges[(src, dst)] = edge
        ty = edge["type"]           # a search only walks edges of its kinds: a 'locz' edge (every non-key-frame adds one) leaves
        for k in [k for k in self._searches if ty in k[2]]:      # the odometry / loop neighbourhoods of the key-frames alone
            del self._searches[k]
        if ty != "locz":
            self._near.clear()
        self.adj.setdefault(src, []).append((dst, edge))
        self.adj.setdefault(dst, []).append((src, edge))

    def _neighbors(self, tok: int, kinds=None) -> List[int]:
        """PoseGraph.get_neighbor_tokens, optionally filtered by edge type: the other ends of the scan's edges, in the
        order the edges entered the graph (the reference walks its edge dict; an adjacency list per scan gives the same
        order)"""
        return [n for n, e in self.adj.get(tok, ()) if kinds is None or e["type"] in kinds]

    def graph_search(self, tok: int, level: int = 5, kinds=("odom", "loop"), max_k: Optional[int] = 16) -> List[int]:
        """PoseGraph.graph_search (pose_graph.py:513-542)"""
        key = (tok, level, tuple(kinds), max_k)
        hit = self._searches.get(key)          # the partner search and the key-frame rule of one scan ask the same question
        if hit is not None:
            return list(hit)
        found: Dict[int, None] = {}
        bfs = [(level, tok)]
        while bfs and (max_k is None or len(found) < max_k):
            rem, t = bfs.pop(0)
            if t in found:
                continue
            found[t] = None
            if rem <= 0:
                continue
            bfs += [(rem - 1, n) for n in self._neighbors(t, kinds)]
        if len(self._searches) > 64:
            self._searches.clear()
        self._searches[key] = list(found)
        return list(found)

    def shortest_path_length(self, src: int, dst: int, kinds=("odom", "loop"), infinity_length: int = 50) -> int:
        """PoseGraph.shortest_path_length (pose_graph.py:544-563)"""
        if src == dst:
            return 0
        vis, bfs = set(), [(0, src)]
        while bfs:
            d, t = bfs.pop(0)
            if t == dst:
                return d
            if t in vis:
                continue
            vis.add(t)
            if d >= infinity_length:
                continue
            bfs += [(d + 1, n) for n in self._neighbors(t, kinds)]
        return infinity_length

    def repair_coor_sys(self) -> None:
        """PoseGraph.repair_coor_sys (pose_graph.py:844-864): connected scans share the smallest coordinate-system id of
        their component.  (The reference takes the unvisited scan with the smallest id as the seed of every flood fill and
        removes visited scans from a list: quadratic in the scans; one pass over the components gives the same labels.)"""
        seen = set()
        for seed in self.type:
            if seed in seen:
                continue
            comp, stack = [], [seed]
            seen.add(seed)
            while stack:
                t = stack.pop()
                comp.append(t)
                for n, _ in self.adj.get(t, ()):
                    if n not in seen:
                        seen.add(n)
                        stack.append(n)
            coor = min(self.coor[t] for t in comp)
            for t in comp:
                self.coor[t] = coor

    def map_tokens(self, tok: int, level: int = 5, max_dist: Optional[float] = 20.0) -> List[int]:
        """The scans of PoseGraph.global_map_query_graph (pose_graph.py:491-496): key-frames of the graph neighbourhood,
        closer than `max_dist` to the centre scan"""
        c = self.poses[tok][:3, 3:]
        toks = [t for t in self.graph_search(tok, level) if self.type[t] != "non-keyframe"]
        if max_dist is not None:
            toks = [t for t in toks if torch.norm(self.poses[t][:3, 3:] - c, p=2, dim=0).item() < max_dist]
        return toks

    def _store_put(self, tok: int, desc: torch.Tensor) -> None:
        if self.store is None:
            self.store = MapTileStore(self.device, channels=desc.shape[0], points=desc.shape[1])
        self.store.put(tok, desc)

    def _rec(self, call: dict) -> None:
        if self.keep_log:
            self.log.append(call)

    def _tile(self, toks: List[int], centre: torch.Tensor):
        self._rec(dict(kind="tile", tokens=list(toks)))
        return self.store.tile(toks, [self.poses[t] for t in toks], centre)

    def _register(self, src: torch.Tensor, dst: torch.Tensor, num_sample, what: str, src_tok, dst_tok):
        R, T, conf, rmse = self.decoder.registration_forward(src, dst, num_sample=num_sample)
        SE3 = PoseTool.SE3(R.cpu(), T.cpu())
        self._rec(dict(kind="reg", what=what, src=src_tok, dst=dst_tok, SE3=SE3.clone(), rmse=float(rmse),
                             cols=(src.shape[-1], dst.shape[-1])))
        return SE3, simvec_to_num(conf), float(rmse)

    def _information(self, src: int, dst: int, SE3: torch.Tensor, fallback=None) -> torch.Tensor:
        """calculate_information_matrix_from_pcd of two scans of the graph (utils.py:60-113); without the clouds -- the
        sharded path gathers descriptors and edge rows only -- the row's matrix (odometry edges: exactly this matrix when
        the partner is the predecessor) or, failing that, the identity"""
        if src in self.pcd and dst in self.pcd:
            Rt = torch.cat([SE3[:3, :3].reshape(9), SE3[:3, 3].reshape(3)]).to(self.device)
            G = ops.information_matrix(self.pcd[src], self.pcd[dst], Rt, 1.0).cpu()
            self._rec(dict(kind="info", src=src, dst=dst, G=G.clone()))
            return G
        return fallback.clone() if fallback is not None else torch.eye(6)

    def add_loop_edge(self, src: int, dst: int, SE3: torch.Tensor, information=None, confidence: float = 1.0, rmse: float = 0.0):
        self._add_edge(src, dst, dict(src=src, dst=dst, SE3=SE3.clone(), type="loop", information=information,
                                      confidence=confidence, rmse=rmse))

    def optimise(self, tokens: Optional[List[int]] = None) -> None:
        """PoseGraph.optim (pose_graph.py:565-658) on the key-frames (all, or `tokens`: bench.py's window) and the
        non-'locz' edges between them; afterwards the non-key-frames follow their key-frames"""
        toks = [t for t in (tokens if tokens is not None else self.keyframes) if self.type.get(t) != "non-keyframe"]
        if len(toks) < 2:
            return
        nodes = {t: self.poses[t].double().numpy() for t in toks}
        es = [(a, b, e["SE3"].double().numpy(),
               np.asarray(e["information"] if e["information"] is not None else np.eye(6), dtype=np.float64))
              for (a, b), e in self.edges.items() if e["type"] != "locz" and a in nodes and b in nodes]
        base = min(self.type) if tokens is None else min(toks)
        self._rec(dict(kind="optim", tokens=list(toks), edges=[(a, b, self.edges[(a, b)]["type"]) for a, b, _, _ in es]))
        self.stats["optimisations"] += 1
        if self.optimiser is not None:
            refined = self.optimiser(nodes, es, base)
        else:
            refined, _ = optimize_pose_graph(nodes, es, base_token=base)
        if not refined:
            return
        for t, P in refined.items():
            self.poses[t] = torch.from_numpy(np.asarray(P, dtype=np.float32))
        self._moved()
        # "Adjust non-keyframes" (pose_graph.py:630-656): breadth first from the reference node, a scan that was not
        # optimised takes the pose of the neighbour it is first reached from times the edge between them
        todo = {t for t in self.type if t not in refined}
        bfs, vis = [base], set()
        while bfs and todo:
            t = bfs.pop(0)
            if t in vis:
                continue
            vis.add(t)
            for n in self._neighbors(t):
                if n in todo and (t, n) in self.edges:
                    self.poses[n] = se3_mul(self.poses[t], self.edges[(t, n)]["SE3"])
                    todo.discard(n)
                if n not in vis:
                    bfs.append(n)

    def _neighbourhood(self, tok: int):
        """The key-frames of `graph_search(tok)` and their positions, stacked as the partner search (n,3,1) and the key-frame
        rule (n,3) stack them.  Both ask for the same key-frame scan after scan until the next key-frame arrives, so the
        answer is kept while no odometry / loop edge is added and no key-frame pose changes (`_moved`)."""
        hit = self._near.get(tok)
        if hit is None:
            near = [t for t in self.graph_search(tok) if self.type[t] != "non-keyframe"]
            hit = self._near[tok] = (near, torch.stack([self.poses[t][:3, 3:] for t in near], dim=0),
                                     torch.stack([self.poses[t][:3, 3] for t in near]))
        return hit

    def _moved(self) -> None:
        """a key-frame's pose changed (scan-to-map refinement, optimiser, an upload placed on the cloud's graph)"""
        self._near.clear()

    # -- OdometryThread ------------------------------------------------------------------------------------------------
    def _odometry_candidates(self) -> List[int]:
        """search_candidates (odometry.py:76-101): the key-frames of the last key-frame's graph neighbourhood, nearest to
        the last known pose first"""
        if not self.type or self.last_known_keyframe is None or self.last_known_anyframe is None:
            return []
        last = self.poses[self.last_known_anyframe]
        near, stack31, _ = self._neighbourhood(self.last_known_keyframe)
        if all((t >> 16) == self.agent_id for t in near):
            kfs = near
        else:
            kfs = [t for t in near if (t >> 16) == self.agent_id]
            stack31 = torch.stack([self.poses[t][:3, 3:] for t in kfs], dim=0)
        d = torch.norm(stack31 - last[:3, 3:], p=2, dim=1)
        _, idx = torch.topk(d, dim=0, k=min(len(kfs), self.args["odometer_candidates_num"]), largest=False)
        return [kfs[i] for i in idx.flatten().tolist()]

    # -- MappingThread -------------------------------------------------------------------------------------------------
    def _valid_check(self, tok: int, edge: dict) -> str:
        a = self.args
        if edge["confidence"] < a["edge_confidence_drop"] or edge["rmse"] > a["edge_rmse_drop"]:
            self.drop_bag.append((tok, edge))
            if len(self.drop_bag) >= a["max_continuous_drop_scan"]:
                if a["continuous_drop_scan_strategy"] != "recover":
                    # 'break' (mapping.py:65-74) adds the scan as a vertex and then falls through to process(), which adds
                    # it again: the reference's own assertion (pose_graph.py:176) ends that run
                    raise NotImplementedError("continuous_drop_scan_strategy 'break' does not survive in the reference either")
                self.drop_bag.clear()
                return ACPT
            return DROP
        self.drop_bag.clear()
        return ACPT

    def _keyframe_check(self, tok: int, edge: dict) -> str:
        a = self.args
        if self.dist_auto:
            rmse_ratio = min(edge["rmse"] / a["edge_rmse_drop"], 1.0)
            self.dist_ratio = max(min(0.90 * self.dist_ratio + 0.10 * ((1 - rmse_ratio) ** 2) * 2.0, 2.0), 0.0)
            self.cur_kf_dist = max(self.kf_dist0 * self.dist_ratio, 1.0)
        old = edge["src"]
        self.poses[tok] = se3_mul(self.poses[old], edge["SE3"])
        self._new_coor = self.coor[old]
        self.last_known_keyframe = old
        if self.cur_kf_dist >= 0:
            d = self._neighbourhood(old)[2] - self.poses[tok][:3, 3].unsqueeze(0)
            if float(torch.norm(d, p=2, dim=1).min()) < self.cur_kf_dist:
                return DIST
        return ACPT

    def _scan_to_map(self, tok: int, edge: dict) -> dict:
        if not self.args["enable_s2m_adjust"]:
            return edge
        old = edge["src"]
        toks = self.map_tokens(old)
        if self.keep_log:
            self.tiles.append(list(toks))
        tile, owner = self._tile(toks, self.poses[old])
        src = tile[:, (owner != tok).to(self.device)]                    # "drop same descriptors from map" (mapping.py:146)
        SE3, conf, rmse = self._register(src, self.desc[tok], self.args["registration_sample_mapping"], "s2m",
                                         [t for t in toks if t != tok], tok)
        self.stats["s2m"] += 1
        return dict(src=old, dst=tok, SE3=se3_inv(SE3), type="odom",
                    information=self._information(old, tok, SE3, edge["information"]), confidence=conf, rmse=rmse)

    # -- LoopThread ----------------------------------------------------------------------------------------------------
    def _loop_detection(self, tok: int, targets: str) -> List[int]:
        """loop_closure_detection (loop_closure.py:92-186)"""
        a = self.args
        cand = [t for t, ty in self.type.items() if ty == "full" and t in self.desc]
        if targets == "self":
            cand = [t for t in cand if (t >> 16) == (tok >> 16)]
        elif targets == "others":
            cand = [t for t in cand if (t >> 16) != (tok >> 16)]
        elif targets != "all":
            raise RuntimeError(f"add_loop_closure received an unknown arg value: targets = {targets}")
        if not cand:
            return []
        trust = a["loop_detection_trust_range"]
        zone1 = set(self.graph_search(tok, trust - 1, max_k=None))
        zone2 = set(self.graph_search(tok, int(trust * 10), max_k=None))
        new = self.poses[tok]
        # (loop_detection_gnss_distance: a ScanPack without a GNSS fix sits at the origin, loop_closure.py:120-123 then
        # keeps every candidate -- nothing on this path carries a fix)
        if a["loop_detection_pred_distance"] > 0:
            off = torch.stack([(self.poses[t] - new)[:2, 3:] for t in cand], dim=0)
            other = torch.tensor([self.coor[t] != self.coor[tok] for t in cand])    # poses of another system say nothing
            keep = (torch.norm(off, p=2, dim=1).squeeze(-1) <= a["loop_detection_pred_distance"]) | other
            cand = [t for t, m in zip(cand, keep) if m]
        if not cand:
            return []
        valid = []
        for t in cand:
            if t in zone1 or t == tok:                                  # trusted zone (too close) or identical
                continue
            if (t >> 16) == (tok >> 16) and t in zone2:                 # trusted zone (turning a corner)
                dR, dT = PoseTool.Rt(se3_mul(se3_inv(self.poses[t]), new))
                if PoseTool.rotation_angle(dR) * 180 / torch.pi < a["loop_detection_rotation_min"] \
                        or torch.norm(dT) < a["loop_detection_translation_min"]:
                    continue
                if self.last_loop_token != -1:
                    _, gap = PoseTool.Rt(se3_mul(se3_inv(self.poses[self.last_loop_token]), new))
                    if torch.norm(gap) < a["loop_detection_transaction_gap"]:
                        continue
            valid.append(t)
        if not valid:
            return []
        src = torch.stack([self.desc[t] for t in valid], dim=0)
        dst = self.desc[tok].unsqueeze(0).expand(len(valid), -1, -1)
        prob = self.decoder.loop_detection_forward(src, dst).cpu()
        self.stats["loop_batches"] += 1
        self._rec(dict(kind="loop", src_tokens=list(valid), dst=tok, prob=prob.clone()))
        # (torch.topk raises when there are fewer candidates than loop_detection_candidates_num; the shipped configs ask
        # for one)
        top, idx = torch.topk(prob, k=min(a["loop_detection_candidates_num"], len(valid)))
        return [valid[i] for i, p in zip(idx.tolist(), top) if p > a["loop_detection_prob_acpt_threshold"]]

    def _loop_registration(self, tok: int, scans: List[int]) -> List[dict]:
        """loop_closure_registration (loop_closure.py:188-258)"""
        out = []
        for prev in scans:
            ptoks, ntoks = self.map_tokens(prev), self.map_tokens(tok)
            self._rec(dict(kind="tile", tokens=list(ptoks)))      # (the reference builds both tiles, then filters
            self._rec(dict(kind="tile", tokens=list(ntoks)))      # their columns; here the token lists are filtered)
            overlap = list(set(ptoks) & set(ntoks))
            if overlap:
                src_t, dst_t = self.poses[prev][:3, 3:], self.poses[tok][:3, 3:]
                ot = torch.cat([self.poses[t][:3, 3:] for t in overlap], dim=1)
                to_prev = torch.norm(ot - src_t, p=2, dim=0) < torch.norm(ot - dst_t, p=2, dim=0)
                o2prev = {t for t, m in zip(overlap, to_prev) if m}
                o2new = set(overlap) - o2prev
                ptoks = [t for t in ptoks if t not in o2new]
                ntoks = [t for t in ntoks if t not in o2prev]
            assert not (set(ptoks) & set(ntoks)) and ptoks and ntoks
            pmap, _ = self.store.tile(ptoks, [self.poses[t] for t in ptoks], self.poses[prev])
            nmap, _ = self.store.tile(ntoks, [self.poses[t] for t in ntoks], self.poses[tok])
            SE3, conf, rmse = self._register(pmap, nmap, self.args["registration_sample_loop"], "loop", list(ptoks), list(ntoks))
            self.stats["loop_registrations"] += 1
            out.append(dict(src=prev, dst=tok, SE3=se3_inv(SE3), type="loop", information=self._information(prev, tok, SE3),
                            confidence=conf, rmse=rmse))
        return out

    def _loop_verification(self, edges: List[dict]) -> List[dict]:
        """loop_closure_verification (loop_closure.py:260-292)"""
        ok = []
        for e in edges:
            if e["confidence"] < self.args["loop_detection_confidence_acpt_threshold"]:
                continue
            dist = self.shortest_path_length(e["src"], e["dst"], infinity_length=5000)
            if dist < 5000:
                delta = se3_mul(se3_inv(se3_mul(self.poses[e["src"]], e["SE3"])), self.poses[e["dst"]])
                dR, dT = PoseTool.Rt(delta)
                if torch.norm(dT).item() / (TRANS_STD * sqrt(dist)) > 3 and dist < 100:
                    continue
                if PoseTool.rotation_angle(dR) * 180 / torch.pi / (ROT_STD * sqrt(dist)) > 3:
                    continue
            ok.append(e)
        return ok

    def global_optimization(self, forced: bool = False) -> bool:
        """loop_closure.py:294-307"""
        a = self.args
        if not a["enable_loop_closure"]:
            return False
        if not forced and (not a["enable_global_optimization"]
                           or self.key_frame_num - self.last_optim_pose_num < a["global_optimization_gap"]
                           or not self.required_optim):
            return False
        self.optimise()
        self.last_optim_pose_num, self.required_optim = self.key_frame_num, False
        return True

    def loop_closure(self, tok: int, targets: Optional[str] = None) -> List[dict]:
        """LoopThread.process (loop_closure.py:56-90)"""
        a = self.args
        if not a["enable_loop_closure"] or self.key_frame_num - self.last_loop_pose_num <= a["loop_detection_gap"]:
            return []
        valid = self._loop_verification(self._loop_registration(tok, self._loop_detection(tok, targets or self.loop_targets)))
        if valid:
            self.required_optim = True
            for e in valid:
                self._add_edge(e["src"], e["dst"], e)
            self.stats["loop_edges"] += len(valid)
            self.last_loop_pose_num, self.last_loop_token = self.key_frame_num, tok
            self.global_optimization(forced=False)
            if (targets or self.loop_targets) in ("all", "others"):
                self.repair_coor_sys()
        return valid

    # -- SlamSystem.step -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def push(self, tok: int, desc: torch.Tensor, row: Optional[torch.Tensor] = None, pcd: Optional[torch.Tensor] = None) -> str:
        """One scan in sequence order: desc (131,S) on the device; row = its gathered edge row (host): the registration of
        its predecessor (source) against it, or None: the odometry registration runs here; pcd (3,N) metres on the device
        or None.  Returns the reference's exit code ('acpt' / 'drop' / 'dist')."""
        self.n_frames += 1
        if pcd is not None:
            self.pcd[tok] = pcd
        cands = self._odometry_candidates()
        if not cands:                                   # first scan of the graph (core.py:383-388)
            self.poses[tok], self.desc[tok] = torch.eye(4), desc
            self._add_vertex(tok, "full")
            self._store_put(tok, desc)
            self.last_known_anyframe = self.last_known_keyframe = self.chain_base = tok
            self.codes.append(ACPT)
            return ACPT
        kf = cands[0]
        self.desc[tok] = desc
        info = row[ops.RES_HDR:ops.RES_HDR + 36].view(6, 6).clone() if row is not None else None
        if row is not None and self.chain_ok and not self.exact_odometry and kf == self.chain_base:
            v = row[:17].tolist()                       # R row-major, T, rmse, -, -, -, confidence (ops.RES_HDR layout)
            step = se3_inv([v[0:3] + [v[9]], v[3:6] + [v[10]], v[6:9] + [v[11]], [0.0, 0.0, 0.0, 1.0]])   # edge.SE3 = registration^-1 (odometry.py:119)
            rel, conf30, rmse = se3_mul(self.since_kf, step), v[16], v[12]                 # key-frame -> predecessor -> this scan
        else:                                           # the reference's own edge: this scan against the key-frame
            SE3, conf30, rmse = self._register(self.desc[kf], desc, self.args["registration_sample_odometer"], "odom", kf, tok)
            rel, info = se3_inv(SE3), self._information(kf, tok, SE3, info)
            self.stats["re_registrations"] += row is not None
        edge = dict(src=kf, dst=tok, SE3=rel, type="odom", information=info, confidence=conf30, rmse=rmse)
        code = self._valid_check(tok, edge)
        if code != ACPT:
            self.stats["dropped"] += 1
            self.chain_ok = False                       # the next scan's consecutive edge hangs on a dropped scan
            self.pcd.pop(tok, None), self.desc.pop(tok, None)
            self.codes.append(code)
            return code
        self.last_known_keyframe = edge["src"]
        code = self._keyframe_check(tok, edge)
        if code != ACPT:
            self._add_vertex(tok, "non-keyframe", self._new_coor)
            self.last_known_anyframe = tok
            self._add_edge(edge["src"], tok, dict(edge, type="locz"))
            self.since_kf, self.chain_base, self.chain_ok = edge["SE3"].clone(), edge["src"], True
            del self.desc[tok]                          # ScanPack.nonkeyframe(): never registered against again
            self.pcd.pop(tok, None)
            self.codes.append(code)
            return code
        self._add_vertex(tok, "full", self._new_coor)
        self._store_put(tok, self.desc[tok])
        self.last_known_anyframe = self.last_known_keyframe = tok
        self._add_edge(edge["src"], tok, dict(edge))
        new = self._scan_to_map(tok, edge)
        if new["rmse"] <= self.args["edge_rmse_drop"] or new["rmse"] <= edge["rmse"]:     # mapping.py:193-201
            self.poses[tok] = se3_mul(self.poses[new["src"]], new["SE3"])
            self._moved()
            self.edges[(edge["src"], tok)].update(SE3=new["SE3"], confidence=new["confidence"],
                                                  information=new["information"], rmse=new["rmse"])
        self.since_kf, self.chain_base, self.chain_ok = torch.eye(4), tok, True
        self.loop_closure(tok)
        self.last_known_anyframe = tok
        if self.optimize_every and self.stats["keyframes"] % self.optimize_every == 0:
            window = self.keyframes[-self.optimize_every:]
            a = window[0]
            if (a, tok) not in self.edges and (tok, a) not in self.edges:
                self.add_loop_edge(a, tok, se3_mul(se3_inv(self.poses[a]), self.poses[tok]), torch.eye(6) * 10.0)
            self.optimise(window)
        self.codes.append(ACPT)
        return ACPT

    def step(self, desc: torch.Tensor, pcd: Optional[torch.Tensor] = None) -> Tuple[int, str]:
        """SlamSystem.step after the extraction (core.py:382-407) for one agent: token = (agent << 16) + frame number
        counted from zero (core.py:361, pose_graph.py:39).  -> (token, exit code)"""
        tok = (self.agent_id << 16) + self.n_frames
        return tok, self.push(tok, desc, None, pcd)

    # -- multi-agent: what an agent uploads and what the cloud does with it ---------------------------------------------------
    def upload_message(self, tok: int) -> dict:
        """The UPLOAD_SCAN message of SlamSystem.step (core.py:409-422) for the key-frame just accepted: the scan, its
        odometry edge and its other edges (the loop edges this step found)"""
        scan = dict(token=tok, agent_id=tok >> 16, timestep=tok & 0xFFFF, type=self.type[tok], key_points=self.desc[tok],
                    full_pcd=self.pcd.get(tok), SE3_pred=self.poses[tok].clone(), coor_sys=self.coor[tok])
        odom = next((e for (a, b), e in self.edges.items() if b == tok and e["type"] == "odom"), None)
        others = [e for n, e in self.adj.get(tok, ()) if e is not odom]
        pack = lambda e: None if e is None else {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in e.items()}
        return dict(new_scan=scan, odometer_edge=pack(odom), neighbor_edges=[pack(e) for e in others])

    @torch.no_grad()
    def cloud_step(self, scan: dict, odom_edge: Optional[dict], neighbor_edges: List[dict]) -> List[dict]:
        """CloudSystem.step (core.py:466-514): the uploaded key-frame joins the cloud's graph at the end of its odometry edge,
        its other edges follow, then the loop closure against the OTHER agents' key-frames (whose accepted edges merge the
        coordinate systems).  -> the loop edges it added."""
        assert scan["type"] == "full"
        tok = scan["token"]
        dev = self.device
        self.n_frames += 1
        self.desc[tok] = scan["key_points"].to(dev)
        if scan.get("full_pcd") is not None:
            self.pcd[tok] = scan["full_pcd"].to(dev)
        self.poses[tok] = scan["SE3_pred"].cpu().clone()
        self._add_vertex(tok, "full", scan["coor_sys"])
        self._store_put(tok, self.desc[tok])
        if odom_edge is not None:
            assert tok in (odom_edge["src"], odom_edge["dst"])
            if tok == odom_edge["src"]:
                self.poses[tok] = se3_mul(self.poses[odom_edge["dst"]], se3_inv(odom_edge["SE3"]))
                self.coor[tok] = self.coor[odom_edge["dst"]]
            else:
                self.poses[tok] = se3_mul(self.poses[odom_edge["src"]], odom_edge["SE3"])
                self.coor[tok] = self.coor[odom_edge["src"]]
            self._add_edge(odom_edge["src"], odom_edge["dst"], dict(odom_edge))
        for e in neighbor_edges:
            self._add_edge(e["src"], e["dst"], dict(e))
        # "for those agents which did not update their pose graph in time" (core.py:487-503): a scan that still carries its
        # agent's old coordinate system is placed from its neighbours in the cloud's graph
        base = min((t for t in self.type if (t >> 16) == (tok >> 16)), key=lambda t: t & 0xFFFF)
        if self.coor[base] != self.coor[tok]:
            for n in self._neighbors(tok):
                if (n, tok) in self.edges:
                    self.poses[tok], self.coor[tok] = se3_mul(self.poses[n], self.edges[(n, tok)]["SE3"]), self.coor[n]
        self._moved()
        return self.loop_closure(tok, "others")

    @torch.no_grad()
    def consume(self, desc: torch.Tensor, table: torch.Tensor, pcd: Optional[torch.Tensor] = None) -> float:
        """One gathered step: desc (n,131,S), table (n,EDGE_FLOATS) on the device (pcd (n,3,N) metres, if gathered), frames
        in sequence order.  Returns the wall time in ms (device work included: the caller's results are only final after
        it)."""
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        head = table[:, :ops.RES_HDR + 36].cpu()       # the one download: 56 floats per frame
        base = self.n_frames
        for g in range(desc.shape[0]):
            self.push(base + g, desc[g], head[g], None if pcd is None else pcd[g])
        torch.cuda.synchronize(self.device)
        return (time.perf_counter() - t0) * 1e3
