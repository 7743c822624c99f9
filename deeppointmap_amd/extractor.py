"""The extractor stage of the reference's multi-thread mode on the batched GPU path (minimal form).

`SlamSystem.MT_ExtractorThread` (reference system/core.py:134-185) is the one place where the reference batches the
encoder: it drains up to EXTRACTOR_BATCHSIZE = 32 pre-processed scans of equal padded length from its input queue,
concatenates them to (B,3,N) / (B,N) and calls `ExtractionThread.process(point_cloud=..., padding_mask=...)`
(system/modules/odometry.py:36-54), which returns the unified descriptors (B,131,256).

`MTExtractor.process` honours that contract on the HIP path -- one launch chain for the whole batch, the descriptor
tensor written by the encoder's last kernel -- so a reference-side `self.extraction_thread = MTExtractor(...)` makes
`--multi_thread` runs batch the encoder as designed.  `MTExtractor.run` is the thread body itself with a software
pipeline on top: while batch i is in its feature stage the sampling (geometry) stage of batch i+1 is already running
on a side stream, which is what HotPath.submit does for bench.py.
"""
from __future__ import annotations

from queue import Empty
from typing import Callable, List, Optional

import torch

from .encoder import Encoder
from .pipeline import _tensors


class MTExtractor:
    EXTRACTOR_BATCHSIZE = 32  # system/core.py:31

    def __init__(self, encoder: Encoder, coor_scale: float = 60.0):
        self.encoder = encoder
        self.coor_scale = float(coor_scale)
        self._side: Optional[torch.cuda.Stream] = None

    # -- ExtractionThread.process (odometry.py:36-54) ----------------------------------------------------------------
    @torch.no_grad()
    def process(self, point_cloud: torch.Tensor, padding_mask: torch.Tensor) -> torch.Tensor:
        """(B,3+,N) normalised scans + (B,N) bool padding (CPU or GPU) -> descriptors (B,131,256) on the GPU:
        rows 0-127 feature, 128-130 key-point coordinates in metres."""
        return self.encoder(point_cloud, padding_mask, descriptor_scale=self.coor_scale)

    # -- the queue contract of MT_ExtractorThread (core.py:139-160) -----------------------------------------------------
    def drain(self, queue_in, is_exit: Callable[[object], bool]):
        """Blocks for one item, then takes what is already queued, up to EXTRACTOR_BATCHSIZE.  Returns
        (scans, controls): the data items in arrival order and the control items (exit codes) met on the way."""
        items = [queue_in.get()]
        while len(items) < self.EXTRACTOR_BATCHSIZE:
            try:
                items.append(queue_in.get_nowait())
            except Empty:
                break
        scans = [it for it in items if not is_exit(it)]
        return scans, [it for it in items if is_exit(it)]

    @staticmethod
    def _collate(scans: List[tuple]):
        """items are (time_ms, point_cloud (1,C,N), R, T, padding_mask (1,N), original_scan) (core.py:152).  The reference
        concatenates and so needs its dataloader to have padded every scan to one length; scans of different lengths are
        padded here (zeros, masked) -- the encoder's output does not depend on masked points."""
        n = max(s[1].shape[2] for s in scans)
        if all(s[1].shape[2] == n for s in scans):
            return torch.cat([s[1] for s in scans], dim=0), torch.cat([s[4] for s in scans], dim=0)
        pts = torch.zeros(len(scans), scans[0][1].shape[1], n, dtype=scans[0][1].dtype, device=scans[0][1].device)
        pad = torch.ones(len(scans), n, dtype=torch.bool, device=scans[0][4].device)
        for i, s in enumerate(scans):
            pts[i, :, :s[1].shape[2]], pad[i, :s[1].shape[2]] = s[1][0], s[4][0]
        return pts, pad

    @torch.no_grad()
    def run(self, queue_in, queue_out, make_scan: Callable, is_exit: Callable[[object], bool], is_final: Callable[[object], bool],
            to_host: bool = True):
        """Thread body: batches from queue_in -> make_scan(item, descriptors (131,256)) objects on queue_out (descriptors on
        the CPU as ScanPack holds them, or -- to_host=False -- left on the device, finished, for a consumer on another stream:
        it must `record_stream` them); exit codes
        are forwarded where the reference forwards them (ahead of the scans of the batch they were drained with,
        core.py:147-151); returns after the final exit code.  Two batches overlap: the sampling stage of the batch just
        drained runs on a side stream while the previous batch goes through its feature stage."""
        dev = self.encoder.device
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)

        def finish(batch):
            scans_p, pts_p, pad_p, pre_p, ready_p = batch
            main.wait_event(ready_p)
            desc = self.encoder(pts_p, pad_p, presampled=pre_p, descriptor_scale=self.coor_scale)
            if to_host:
                desc = desc.cpu()
            else:
                main.synchronize()
            for item, d in zip(scans_p, desc):
                queue_out.put(make_scan(item, d))

        pending = None
        while True:
            scans, controls = self.drain(queue_in, is_exit)
            nxt = None
            if scans:
                pts, pad = self._collate(scans)
                pts, pad = pts.to(dev, non_blocking=True), pad.to(dev, non_blocking=True)
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    pre = self.encoder.presample(pts, pad)
                    for t in _tensors(pre):
                        t.record_stream(main)
                    ready = self._side.record_event()
                nxt = (scans, pts, pad, pre, ready)
            if pending is not None:
                finish(pending)
            for c in controls:
                queue_out.put(c)
            pending = nxt
            if any(is_final(c) for c in controls):
                break
        if pending is not None:
            finish(pending)
