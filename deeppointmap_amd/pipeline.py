"""The hot path as one call: encode a batch of scans, register consecutive frames, build the
information matrix of each edge -- what SlamSystem.step does per frame through
ExtractionThread.process + OdometryThread.odometry (reference system/core.py:369-393,
system/modules/odometry.py:36-54,103-127), minus the pose-graph bookkeeping.

Used by bench.py, the smoke test and the parity tests; frame sharding across ranks lives in
shard.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch

from . import ops
from .decoder import Decoder
from .encoder import Encoder
from .registration import make_descriptors

EDGE_FLOATS = 56  # per edge: 20-float registration header (R, T, rmse, n_corr, n_inlier, iters, conf30, ...) + 6x6 information


@dataclass
class Edge:
    """One odometry edge (what PoseGraph_Edge receives, odometry.py:119-125)."""
    src: int
    dst: int
    R: torch.Tensor           # (3,3) device
    T: torch.Tensor           # (3,1) device, metres
    conf: torch.Tensor        # (n_inlier,) device
    rmse: float
    information: Optional[torch.Tensor]  # (6,6) device


class HotPath:
    def __init__(self, encoder: Encoder, decoder: Decoder, coor_scale: float = 60.0, num_sample=0.5):
        self.encoder, self.decoder = encoder, decoder
        self.coor_scale, self.num_sample = float(coor_scale), num_sample
        self._side = None      # side HIP stream for the software pipeline (submit / flush)
        self._pending = None

    @torch.no_grad()
    def extract(self, points: torch.Tensor, padding: torch.Tensor, presampled=None) -> torch.Tensor:
        """(F,3,N) normalised scans -> unified descriptors (F,131,256): rows 0-127 feature, 128-130 xyz in metres."""
        coor, fea, _ = self.encoder(points, padding, presampled=presampled)
        return make_descriptors(coor, fea, self.coor_scale)

    @torch.no_grad()
    def register(self, desc: torch.Tensor, pcd_m: Optional[torch.Tensor], pairs, table: Optional[torch.Tensor] = None,
                 materialize: bool = True):
        """desc (F,131,S); pcd_m (F,3,N) scans in metres (None: skip the information matrix);
        pairs: list of (src_frame, dst_frame).  All pairs are registered in ONE batched pass.
        Returns (edges, table): table (E, EDGE_FLOATS) is filled on the device by the kernels themselves
        (20-float registration header | 6x6 information) -- it is what a rank ships to rank 0.
        materialize=False skips building Edge objects (no host synchronisation at all)."""
        pairs = list(pairs)
        dev = desc.device
        if table is None:
            table = torch.zeros(len(pairs), EDGE_FLOATS, device=dev, dtype=torch.float32)
        sidx = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev)
        didx = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev)
        res = self.decoder.registration_forward_batch(desc.index_select(0, sidx.long()), desc.index_select(0, didx.long()),
                                                      num_sample=self.num_sample, header_out=table[:, :ops.RES_HDR])
        if pcd_m is not None:
            ops.information_matrix_batched(pcd_m, sidx, didx, table[:, :12], table[:, ops.RES_HDR:])
        edges = []
        if materialize:
            head = table[:, :ops.RES_HDR].cpu()
            for e, (s, d) in enumerate(pairs):
                n_in = int(head[e, 14])
                info = table[e, ops.RES_HDR:].view(6, 6) if pcd_m is not None else None
                edges.append(Edge(s, d, res[e, 0:9].view(3, 3), res[e, 9:12].view(3, 1),
                                  res[e, ops.RES_HDR:ops.RES_HDR + n_in], float(head[e, 12]), info))
        return edges, table

    @torch.no_grad()
    def step(self, points: torch.Tensor, padding: torch.Tensor, pcd_m: Optional[torch.Tensor], materialize: bool = True):
        """One batch: every frame is encoded and registered against its predecessor (frame 0 against
        the last frame of the batch, so a batch of F frames carries exactly F edges)."""
        desc = self.extract(points, padding)
        F = desc.shape[0]
        edges, table = self.register(desc, pcd_m, [((f - 1) % F, f) for f in range(F)], materialize=materialize)
        return desc, edges, table

    # -- streaming mode: two-stage software pipeline over consecutive batches ---------------------------
    @torch.no_grad()
    def submit(self, points: torch.Tensor, padding: torch.Tensor, pcd_m: Optional[torch.Tensor]):
        """Enqueue a batch.  Its input staging + first-level FPS (one CU per frame, latency-bound) start at
        once on a side stream; the rest of the PREVIOUS batch (remaining encoder stages, registration,
        information matrices) is enqueued on the current stream and overlaps with it.  Returns the previous
        batch's (desc, table) or None for the first call; flush() returns the last one.  Inputs must already
        be ready on the device (they are read from the side stream without waiting for the current one)."""
        dev = self.encoder.device
        main = torch.cuda.current_stream(dev)
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(self._side):
            pre = self.encoder.presample(points, padding)
            ready = self._side.record_event()
        for t in pre.values():
            t.record_stream(main)  # produced on the side stream, consumed on the main one
        prev, self._pending = self._pending, (pre, ready, points, padding, pcd_m)
        return self._finish(prev) if prev is not None else None

    @torch.no_grad()
    def flush(self):
        prev, self._pending = self._pending, None
        return self._finish(prev) if prev is not None else None

    def _finish(self, item):
        pre, ready, points, padding, pcd_m = item
        torch.cuda.current_stream(self.encoder.device).wait_event(ready)
        desc = self.extract(points, padding, presampled=pre)
        F = desc.shape[0]
        _, table = self.register(desc, pcd_m, [((f - 1) % F, f) for f in range(F)], materialize=False)
        return desc, table
