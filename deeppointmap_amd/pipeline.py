"""The hot path as one call: encode a batch of scans, register consecutive frames, build the
information matrix of each edge -- what SlamSystem.step does per frame through
ExtractionThread.process + OdometryThread.odometry (reference system/core.py:369-393,
system/modules/odometry.py:36-54,103-127), minus the pose-graph bookkeeping.

Used by bench.py, the smoke test and the parity tests; frame sharding across ranks lives in
shard.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import ops
from .decoder import Decoder
from .encoder import Encoder

EDGE_FLOATS = 56  # per edge: 20-float registration header (R, T, rmse, n_corr, n_inlier, iters, conf30, ...) + 6x6 information


def _tensors(d):
    """every tensor of a (possibly nested) dict / list / tuple"""
    if isinstance(d, torch.Tensor):
        yield d
    elif isinstance(d, dict):
        for v in d.values():
            yield from _tensors(v)
    elif isinstance(d, (list, tuple)):
        for v in d:
            yield from _tensors(v)


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
        self.geometry_levels = None  # FPS levels run by the geometry stage (None = all; measured best on MI355X)
        self.geometry_depth = 2      # geometry launches in flight ahead of the feature stage (one HIP stream each)
        self.geometry_group = 1      # batches whose first-level sampling shares ONE launch
        self.feature_streams = 1     # >1: consecutive batches' feature stages alternate between side streams
        # > 0: the feature stage as TWO pipeline stages -- the downsampling levels below `feature_split` on the caller's stream,
        # the rest (lower levels, upsamplers, descriptors) on a stream of its own, so that the next batch's first level does
        # not wait for this batch's launch-bound tail.  Built because every extra evaluation of levels 3+ costs the pipelined
        # step its whole 0.44 ms (scripts/price_tail.py); bit-identical results -- and SLOWER at every cut (4.88 / 4.58 / 4.39 /
        # 4.89 ms per step at levels 1 / 2 / 3 / 4 against 4.18, with 8 hardware queues; worse with the default 4, where a fifth
        # stream shares a queue with a sampling launch): one more kernel stream on the chip costs more than the shorter chain
        # returns.  Off.
        self.feature_split = 0
        # chain = True: frame 0 of a batch is registered against the frame BEFORE the batch (the previous batch's last
        # frame, or -- several ranks -- the last frame of the rank that owns the preceding block, shard.exchange_halo)
        # instead of the batch's own last frame (the single-GPU ring, which costs the same and needs no hand-over)
        self.chain = False
        self._halo = None      # (descriptor, scan) of the frame before the next batch
        self._no_predecessor = False
        self._side = None      # side HIP streams for the software pipeline (submit / flush)
        self._pending = None
        self._rings = {}
        # Bytes handed to torch's caching allocator as ONE segment before the first pipelined batch.  Four batches are in flight
        # on four streams; a block freed on one stream while another still reads it cannot be reused yet, so the allocator
        # keeps meeting requests nothing cached fits and goes to hipMalloc -- 5-12 ms of host time each, 28 of them over the
        # first 150 steps until ~6.3 GB were reserved (scripts/debug/step_hiccup.py), and a 20-step measurement that catches a
        # burst of them reads 5.3 instead of 4.35 ms per step.  Blocks of one big cached segment are split instead.
        # Default 16 GiB of the chip's 288 (the measured working set of the four batches in flight is ~6.3 GB, spread over the
        # pipeline's seven streams), capped at half of what is free when the pipeline starts: the shipped configuration is the one
        # bench.py measures.  A caller that shares the GPU with other tenants sets 0 (nothing reserved).
        self.reserve_bytes = 16 << 30
        # False: the caller guarantees that a batch's inputs are complete (or ordered by an event the geometry streams already wait
        # for) when submit() is called -- resident scans, inputs staged on a stream of the caller's own that it synchronised with
        # `for g in hot.geometry_streams(): g.wait_event(ev)` -- and the geometry stage does not wait for the caller's stream.
        self.inputs_on_caller_stream = True

    @torch.no_grad()
    def extract(self, points: torch.Tensor, padding: torch.Tensor, presampled=None) -> torch.Tensor:
        """(F,3,N) normalised scans -> unified descriptors (F,131,256): rows 0-127 feature, 128-130 xyz in metres.
        In chain mode the tensor has one more slot (index F) holding the predecessor of frame 0."""
        return self.encoder(points, padding, presampled=presampled, descriptor_scale=self.coor_scale,
                            spare_frames=1 if self.chain else 0)

    def _hand_over(self, desc: torch.Tensor, pcd_m: Optional[torch.Tensor]):
        """chain mode, after extract(): fill slot F of `desc` with the frame before this batch and remember this batch's
        last frame for the next one.  Returns that predecessor's scan (3,N) (None without scans)."""
        import torch.distributed as dist
        from .shard import exchange_halo
        F = desc.shape[0] - 1
        last = (desc[F - 1], pcd_m[F - 1] if pcd_m is not None else None)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        # the very first batch of a chain (on rank 0) has no predecessor: slot F is filled with the batch's own last frame so
        # that the kernels have something to read, and register() marks edge 0 as void (see `_void_first`)
        self._no_predecessor = self._halo is None and (not multi or dist.get_rank() == 0)
        if multi:
            got = exchange_halo(*last)
            if dist.get_rank() == 0:  # what arrives is the end of the whole window: the predecessor for the NEXT step
                use, self._halo = (self._halo if self._halo is not None else last), got
            else:
                use = got
        else:
            use, self._halo = ((self._halo if self._halo is not None else last),
                               (last[0].clone(), last[1].clone() if last[1] is not None else None))
        desc[F].copy_(use[0])
        return use[1]

    @staticmethod
    def _void_first(table: torch.Tensor) -> None:
        """Edge row 0 of a chain's first batch: there is no frame before frame 0, so the row carries no edge -- identity pose,
        rmse +inf, zero correspondences / inliers, zero information.  Consumers skip a row with n_inlier == 0 and infinite rmse
        (Rank0Consumer never reads the row of the first scan of its graph)."""
        row = torch.zeros(table.shape[1], device=table.device, dtype=table.dtype)
        row[0], row[4], row[8], row[12] = 1.0, 1.0, 1.0, float("inf")
        table[0].copy_(row)

    @torch.no_grad()
    def register(self, desc: torch.Tensor, pcd_m: Optional[torch.Tensor], pairs, table: Optional[torch.Tensor] = None,
                 materialize: bool = True, pair_index=None, grids: Optional[torch.Tensor] = None,
                 halo_pcd: Optional[torch.Tensor] = None):
        """desc (F,131,S); pcd_m (F,3,N) scans in metres (None: skip the information matrix);
        pairs: list of (src_frame, dst_frame).  All pairs are registered in ONE batched pass.
        Returns (edges, table): table (E, EDGE_FLOATS) is filled on the device by the kernels themselves
        (20-float registration header | 6x6 information) -- it is what a rank ships to rank 0.
        materialize=False skips building Edge objects (no host synchronisation at all).
        pair_index: (src, dst) int32 device tensors of `pairs` when the caller already holds them;
        grids: ops.information_matrix_grids(pcd_m, dst) built ahead of time (the pose-independent half).
        halo_pcd (chain mode): scan (3,N) of the frame in descriptor slot F, the source of pair 0."""
        pairs = list(pairs)
        dev = desc.device
        if table is None:
            table = torch.empty(len(pairs), EDGE_FLOATS, device=dev, dtype=torch.float32)  # every field is written below
        if pair_index is None:
            pair_index = (torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev),
                          torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev))
        sidx, didx = pair_index[0], pair_index[1]
        if pcd_m is None:
            table[:, ops.RES_HDR:].zero_()
        res = self.decoder.registration_forward_pairs(desc, sidx, didx, num_sample=self.num_sample,
                                                      header_out=table[:, :ops.RES_HDR],
                                                      order=pair_index[2] if len(pair_index) > 2 else None)
        if pcd_m is not None:
            F = pcd_m.shape[0]
            if halo_pcd is None and any(p[0] >= F for p in pairs):
                raise ValueError("chain mode: the frame before this batch came without its scan, so the information matrix of "
                                 "the first edge cannot be built (pass scans for every batch of a chain, or for none)")
            if halo_pcd is not None:
                # the batched search addresses scans of ONE tensor: it runs with the ring's sources (the grids were built
                # for exactly these targets), and pair 0 -- whose source scan is the hand-over frame -- is redone alone
                ring = self._ring_pairs(F, dev)[1]
                ops.information_matrix_batched(pcd_m, ring[0], didx, table[:, :12], table[:, ops.RES_HDR:], grids=grids)
                table[0, ops.RES_HDR:].copy_(ops.information_matrix(halo_pcd.contiguous(), pcd_m[pairs[0][1]], table[0, :12], 1.0).reshape(-1))
            else:
                ops.information_matrix_batched(pcd_m, sidx, didx, table[:, :12], table[:, ops.RES_HDR:], grids=grids)
        edges = []
        if materialize:
            head = table[:, :ops.RES_HDR].cpu()
            for e, (s, d) in enumerate(pairs):
                n_in = int(head[e, 14])
                info = table[e, ops.RES_HDR:].view(6, 6) if pcd_m is not None else None
                edges.append(Edge(s, d, res[e, 0:9].view(3, 3), res[e, 9:12].view(3, 1),
                                  res[e, ops.RES_HDR:ops.RES_HDR + n_in], float(head[e, 12]), info))
        return edges, table

    def _ring_pairs(self, F: int, dev, chain: bool = False):
        """Every frame against its predecessor: the pair list and its device index tensors, built once per batch size.
        Frame 0's predecessor is the last frame of the batch (ring) or, chain=True, slot F of the descriptor tensor."""
        key = (F, str(dev), chain)
        if key not in self._rings:
            pairs = [((f - 1) % F if f or not chain else F, f) for f in range(F)]
            si = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev)
            di = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev)
            self._rings[key] = (pairs, (si, di, torch.cat([si, di])))  # + sources-then-targets, for the decoder's gathers
        return self._rings[key]

    @torch.no_grad()
    def step(self, points: torch.Tensor, padding: torch.Tensor, pcd_m: Optional[torch.Tensor], materialize: bool = True):
        """One batch: every frame is encoded and registered against its predecessor (frame 0 against
        the last frame of the batch, so a batch of F frames carries exactly F edges)."""
        desc = self.extract(points, padding)
        F = points.shape[0]
        halo_pcd = self._hand_over(desc, pcd_m) if self.chain else None
        void = self.chain and self._no_predecessor
        pairs, index = self._ring_pairs(F, desc.device, self.chain)
        edges, table = self.register(desc, pcd_m, pairs, materialize=materialize, pair_index=index, halo_pcd=halo_pcd)
        if void:
            self._void_first(table)
            if edges:
                dev = desc.device
                edges[0] = Edge(pairs[0][0], pairs[0][1], torch.eye(3, device=dev), torch.zeros(3, 1, device=dev),
                                torch.empty(0, device=dev), float("inf"), None)
        return desc[:F], edges, table

    # -- streaming mode: software pipeline over consecutive batches ----------------------------------------
    #   stage G (streams A0/A1, alternating): geometry of batch i -- staging + the FPS chain.  It is a latency
    #            chain on one CU per frame, so TWO batches' geometry passes are kept in flight (`geometry_depth`):
    #            the stage's throughput, not its latency, then bounds the step.
    #   stage F (caller's stream): features of batch i-depth -- kNN, grouped MLPs, GEMMs -> descriptors
    #   stage R (stream B): registration of the batch before that -- decoder + information matrices
    # The stages touch disjoint data, so they overlap on the chip; flush() drains the pipe, so K submits + flush
    # contain exactly K batches of work.  A finished batch is handed out one submit after its registration was enqueued
    # (`_hand_out`), so the caller's stream never waits for work that has only just been queued.
    @torch.no_grad()
    def submit(self, points: torch.Tensor, padding: torch.Tensor, pcd_m: Optional[torch.Tensor]):
        """Enqueue a batch; returns the (desc, table) of an earlier batch once the pipe is full (None while it
        fills).  The geometry stage waits for whatever the caller's stream has enqueued so far, so inputs may come from
        asynchronous copies or kernels on that stream."""
        dev = self.encoder.device
        if self._side is None:
            self._side = dict(geo=[torch.cuda.Stream(device=dev) for _ in range(max(1, self.geometry_depth))],
                              reg=torch.cuda.Stream(device=dev),
                              feat=[torch.cuda.Stream(device=dev) for _ in range(self.feature_streams)]
                              if self.feature_streams > 1 else [],
                              featb=torch.cuda.Stream(device=dev) if self.feature_split else None)
            if self.reserve_bytes:
                # the allocator's cache is per stream: every stream of the pipeline gets its share
                streams = [torch.cuda.current_stream(dev)] + self._side["geo"] + [self._side["reg"]] + self._side["feat"] + \
                    ([self._side["featb"]] if self._side["featb"] is not None else [])
                free = torch.cuda.mem_get_info(dev)[0]
                n = min(int(self.reserve_bytes), free // 2) // len(streams)
                if n >= (1 << 29):
                    for st in streams:
                        with torch.cuda.stream(st):
                            del_me = torch.empty(n, dtype=torch.uint8, device=dev)  # one segment; back into the cache at once
                            del del_me
            self._pending = dict(geo=[], reg=None, n=0, nf=0, hold=[])
        self._pending["hold"].append((points, padding, pcd_m))
        if len(self._pending["hold"]) >= max(1, self.geometry_group):
            self._launch_geometry()
        done = None
        if len(self._pending["geo"]) + len(self._pending["hold"]) > self.geometry_depth * max(1, self.geometry_group) \
                and self._pending["geo"]:
            done = self._advance(self._pending["geo"].pop(0))
        return done

    def _launch_geometry(self):
        """Stage G of the held batches on the next geometry stream: one joint first-level sampling launch, then
        per batch the staging, the lower levels and the search grids."""
        dev = self.encoder.device
        main = torch.cuda.current_stream(dev)
        hold, self._pending["hold"] = self._pending["hold"], []
        sa = self._side["geo"][self._pending["n"] % len(self._side["geo"])]
        self._pending["n"] += 1
        rings = [self._ring_pairs(h[0].shape[0], dev) if h[2] is not None else None for h in hold]  # before the stream switch: a first call copies H2D
        if self.inputs_on_caller_stream:
            # inputs the caller produced asynchronously on its stream (H2D copies, GPU pre-processing).  The feature stage of an
            # earlier batch sits at the head of that stream, so this also makes G(i) wait for F(i - 3) to finish.
            sa.wait_stream(main)
        for points, padding, pcd_m in hold:
            # the caller may drop its tensors as soon as submit() returns: every stream that will read them has to be on
            # record with the allocator, or their memory is handed out again while a stage is still reading it
            # (found by scripts/fuzz_pipeline.py: 4 % of the batches came back with a wrong information matrix)
            for t in (points, padding):
                if t.is_cuda:
                    t.record_stream(sa)
            if pcd_m is not None and pcd_m.is_cuda:
                pcd_m.record_stream(sa)
                pcd_m.record_stream(self._side["reg"])
        with torch.cuda.stream(sa):
            first = self.encoder.sample_first_level([h[0] for h in hold], [h[1] for h in hold]) if len(hold) > 1 else [None]
            for (points, padding, pcd_m), ring, s0 in zip(hold, rings, first):
                pre = self.encoder.presample(points, padding, levels=self.geometry_levels, sampled0=s0)
                ready = sa.record_event()
                scans = None
                if pcd_m is not None:
                    # the information matrix's target grids need no pose: built here, off the registration stream
                    grids = ops.information_matrix_grids(pcd_m, ring[1][1])
                    grids.record_stream(self._side["reg"])
                    scans = (pcd_m, grids, sa.record_event())
                for t in _tensors(pre):
                    t.record_stream(main)  # produced on a geometry stream, consumed on the caller's stream
                self._pending["geo"].append((pre, ready, points, padding, scans))

    def _advance(self, geo):
        """features of `geo` on the caller's stream, then registration of the batch before it on stream B."""
        dev = self.encoder.device
        main = torch.cuda.current_stream(dev)
        sb = self._side["reg"]
        pre, ready, points, padding, pcd_m = geo   # pcd_m: None or (scans, grids, grids_ready)
        halo_pcd = None
        if self._side["feat"]:
            if self.chain:
                raise NotImplementedError("chain mode runs the feature stage on the caller's stream")
            sf = self._side["feat"][self._pending["nf"] % len(self._side["feat"])]
            self._pending["nf"] += 1
            with torch.cuda.stream(sf):
                sf.wait_event(ready)
                for t in _tensors(pre):
                    t.record_stream(sf)
                desc = self.extract(points, padding, presampled=pre)
                desc_ready = sf.record_event()
            desc.record_stream(main)  # handed to the caller (gather, host copies) on its stream
        elif self._side["featb"] is not None and 0 < self.feature_split < self.encoder.downsample_layers:
            fb = self._side["featb"]
            main.wait_event(ready)
            state = self.encoder(points, padding, presampled=pre, stop_level=self.feature_split)
            half = main.record_event()
            for t in _tensors(state):
                t.record_stream(fb)       # made (or handed over) on the caller's stream, read on the second feature stream
            with torch.cuda.stream(fb):
                fb.wait_event(half)
                desc = self.encoder(points, padding, resume=state, descriptor_scale=self.coor_scale,
                                    spare_frames=1 if self.chain else 0)
                desc_ready = fb.record_event()
            desc.record_stream(main)      # handed to the caller (gather, host copies) on its stream
        else:
            main.wait_event(ready)
            desc = self.extract(points, padding, presampled=pre)
            desc_ready = main.record_event()
        desc.record_stream(sb)
        # the registration of the batch before goes onto stream B first ...
        prev, self._pending["reg"] = self._pending["reg"], None
        out = self._register_on_b(prev) if prev is not None else None
        if self.chain:
            # ... then this batch's hand-over, ALSO on stream B: only the registration needs the predecessor frame, so the
            # caller's stream -- the next batch's feature stage -- never waits for the neighbour rank's message.  It is a
            # collective: every rank reaches it once per batch, in batch order.
            with torch.cuda.stream(sb):
                sb.wait_event(desc_ready)
                halo_pcd = self._hand_over(desc, pcd_m[0] if pcd_m is not None else None)
                desc_ready = sb.record_event()
        self._pending["reg"] = (desc, desc_ready, pcd_m, halo_pcd if self.chain else None, self.chain and self._no_predecessor)
        return out

    def _register_on_b(self, reg):
        dev = self.encoder.device
        main = torch.cuda.current_stream(dev)
        sb = self._side["reg"]
        desc, desc_ready, scans, halo_pcd, void = reg
        F = desc.shape[0] - (1 if self.chain else 0)
        pairs, index = self._ring_pairs(F, dev, self.chain)
        with torch.cuda.stream(sb):
            sb.wait_event(desc_ready)
            pcd_m = grids = None
            if scans is not None:
                pcd_m, grids, grids_ready = scans
                sb.wait_event(grids_ready)
            _, table = self.register(desc, pcd_m, pairs, materialize=False, pair_index=index, grids=grids, halo_pcd=halo_pcd)
            if void:
                self._void_first(table)
            done = sb.record_event()
        table.record_stream(main)
        # Results are handed out ONE call later (`_hand_out`): the caller's stream -- which is also the feature stage's --
        # then waits for a registration that finished long ago instead of stalling the next batch's features behind the
        # one that has just been enqueued (the stages would run in lock-step, each step ending with one of them alone).
        out, self._pending["out"] = self._pending.get("out"), (desc[:F], table, done)
        return self._hand_out(out)

    def _hand_out(self, out):
        if out is None:
            return None
        desc, table, done = out
        torch.cuda.current_stream(self.encoder.device).wait_event(done)  # the caller's stream sees finished results
        return desc, table

    def geometry_streams(self):
        """the streams the geometry stage runs on (empty before the first submit): a caller that stages a batch's inputs on a stream
        of its own makes these wait for its event, sets `inputs_on_caller_stream = False`, and the geometry stage no longer waits for
        whatever else sits on the caller's stream"""
        return list(self._side["geo"]) if self._side is not None else []

    @torch.no_grad()
    def flush(self):
        """Drain the pipe: returns the list of (desc, table) still in flight, oldest first."""
        out = []
        if self._side is None:
            return out
        if self._pending["hold"]:
            self._launch_geometry()
        while self._pending["geo"]:
            r = self._advance(self._pending["geo"].pop(0))
            if r is not None:
                out.append(r)
        if self._pending["reg"] is not None:
            reg, self._pending["reg"] = self._pending["reg"], None
            r = self._register_on_b(reg)
            if r is not None:
                out.append(r)
        last, self._pending["out"] = self._pending.get("out"), None
        if last is not None:
            out.append(self._hand_out(last))
        return out
