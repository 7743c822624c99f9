"""The sequential part of the SLAM front/back end that stays on rank 0 after the frame-sharded encode: what
SlamSystem.step does with a new scan once its descriptors and its speculative odometry edge exist (reference
system/core.py:382-407) -- key-frame gating (mapping.py:83-132), scan-to-map refinement of every key-frame against
the tile of its graph neighbourhood (mapping.py:136-170: PoseGraph.global_map_query_graph + registration_forward at
up to 16 * 256 map key-points), and the pose-graph optimisation that follows loop closures (loop_closure.py:296-310).

It consumes what shard.gather_step_results delivers -- descriptors (n,131,S) and edge rows (n,EDGE_FLOATS) in sequence
order -- and is the Amdahl term of the multi-GPU path: its cost per step is reported by bench.py for
N > 1 next to the sharded throughput.  Key-frame selection is by count (every `keyframe_every`-th frame) rather than by
the reference's distance rule: the cost model is what matters here, not the map (with procedural weights the poses
carry no meaning).
"""
from __future__ import annotations

import time
from typing import Dict, List

import numpy as np
import torch

from . import ops
from .maptile import MapTileStore
from .posegraph_optim import optimize_pose_graph


class Rank0Consumer:
    def __init__(self, decoder, device, keyframe_every: int = 8, tile_scans: int = 16, optimize_every: int = 16,
                 num_sample=0.5):
        self.decoder, self.device = decoder, torch.device(device)
        self.keyframe_every, self.tile_scans, self.optimize_every = keyframe_every, tile_scans, optimize_every
        self.num_sample = num_sample
        self.store = MapTileStore(self.device)
        self.poses: Dict[int, torch.Tensor] = {}      # key-frame token -> SE3_pred (4,4) CPU
        self.edges: List[tuple] = []                   # (src, dst, SE3 (4,4) np, information (6,6) np)
        self.keyframes: List[int] = []
        self.n_frames = 0
        self.cur = torch.eye(4)
        self.stats = dict(s2m=0, optimisations=0)

    @torch.no_grad()
    def consume(self, desc: torch.Tensor, table: torch.Tensor) -> float:
        """One gathered step: desc (n,131,S), table (n,EDGE_FLOATS) on the device, frames in sequence order.
        Returns the wall time in ms (device work included: the caller's results are only final after it)."""
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        head = table[:, :ops.RES_HDR + 36].cpu()       # the one download: 56 floats per frame
        for g in range(desc.shape[0]):
            tok = self.n_frames
            self.n_frames += 1
            R, T = head[g, 0:9].view(3, 3), head[g, 9:12].view(3, 1)
            rel = torch.eye(4)
            rel[:3, :3], rel[:3, 3:4] = R, T
            self.cur = self.cur @ torch.linalg.inv(rel)          # edge.SE3 = inverse of the registration (odometry.py:119)
            if tok % self.keyframe_every:
                continue                                         # a non-key-frame: localised, not mapped
            if self.keyframes:
                near = self.keyframes[-self.tile_scans:]
                prev = self.keyframes[-1]
                tile, _ = self.store.tile(near, [self.poses[t] for t in near], self.poses[prev])
                R2, T2, conf, rmse = self.decoder.registration_forward(tile, desc[g], num_sample=self.num_sample)
                ref = torch.eye(4)
                ref[:3, :3], ref[:3, 3:4] = R2.cpu(), T2.cpu()
                self.stats["s2m"] += 1
                if np.isfinite(rmse):
                    self.cur = self.poses[prev] @ torch.linalg.inv(ref)
                self.edges.append((prev, tok, torch.linalg.inv(self.poses[prev]) @ self.cur,
                                   head[g, ops.RES_HDR:ops.RES_HDR + 36].view(6, 6)))
            self.store.put(tok, desc[g])
            self.poses[tok] = self.cur.clone()
            self.keyframes.append(tok)
            if len(self.keyframes) % self.optimize_every == 0 and len(self.edges) >= 2:
                # a loop edge between the window's ends, then the global optimisation it triggers
                a, b = self.keyframes[-self.optimize_every], tok
                self.edges.append((a, b, torch.linalg.inv(self.poses[a]) @ self.poses[b], torch.eye(6) * 10.0))
                window = self.keyframes[-self.optimize_every:]
                nodes = {t: self.poses[t].double().numpy() for t in window}
                es = [(s, d, np.asarray(X, dtype=np.float64), np.asarray(I, dtype=np.float64)) for s, d, X, I in self.edges
                      if s in nodes and d in nodes]
                refined, _ = optimize_pose_graph(nodes, es, base_token=window[0])
                for t, P in refined.items():
                    self.poses[t] = torch.from_numpy(np.asarray(P, dtype=np.float32))
                self.cur = self.poses[tok].clone()
                self.stats["optimisations"] += 1
        torch.cuda.synchronize(self.device)
        return (time.perf_counter() - t0) * 1e3
