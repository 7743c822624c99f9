"""Device-resident map tiles -- the step between the encoder output and scan-to-map / loop registration
(reference PoseGraph.__global_mapping + global_map_query_graph, system/modules/pose_graph.py:373-409,471-511).

The reference keeps every ScanPack on the CPU, re-uploads the selected keyframes' descriptors for every query,
transforms and concatenates them with torch ops, and downloads the tile again.  Here the descriptors stay in HBM
(`MapTileStore`), and one kernel assembles a tile from any ordered selection of scans.  WHICH scans form a tile
(the BFS over the pose graph, the distance cut) stays host logic of the caller.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib, ops


def _pose_rows(SE3: torch.Tensor) -> torch.Tensor:
    """(..., 4, 4) -> (..., 12): R row-major then T."""
    return torch.cat([SE3[..., :3, :3].reshape(*SE3.shape[:-2], 9), SE3[..., :3, 3]], dim=-1)


def assemble_map_tile(key_points: torch.Tensor, poses: torch.Tensor, centering_SE3: torch.Tensor,
                      select: Optional[torch.Tensor] = None) -> torch.Tensor:
    """key_points (n,C,S) fp32 on the GPU, poses (n,4,4) SE3_pred, centering (4,4), select (K,) int32 scan indices
    in tile order (None: all, in order) -> tile (C, K*S) on the GPU."""
    ops._chk(key_points, torch.float32, "key_points")
    dev = key_points.device
    n, C, S = key_points.shape
    P = _pose_rows(poses.to(dtype=torch.float32)).to(dev).contiguous()
    c = _pose_rows(centering_SE3.to(dtype=torch.float32)).to(dev).contiguous()
    if select is not None:
        select = ops._chk(select.to(device=dev, dtype=torch.int32).contiguous(), torch.int32, "select")
    K = n if select is None else select.numel()
    out = torch.empty(C, K * S, device=dev, dtype=torch.float32)
    _lib.check(_lib.load().dpm_map_tile(ops._ptr(key_points), ops._ptr(select), ops._ptr(P), ops._ptr(c), C, S, K,
                                        ops._ptr(out), ops._stream(key_points)), "dpm_map_tile")
    return out


class MapTileStore:
    """Keyframe descriptors resident in HBM, addressed by scan token (what `ScanPack.key_points` +
    `__global_map_cache` are in the reference)."""

    def __init__(self, device, channels: int = 131, points: int = 256, capacity: int = 1024):
        self.device = torch.device(device)
        self._buf = torch.empty(capacity, channels, points, device=self.device, dtype=torch.float32)
        self._slot: Dict[int, int] = {}

    def __len__(self):
        return len(self._slot)

    def put(self, token: int, key_points: torch.Tensor) -> None:
        if token not in self._slot:
            if len(self._slot) == self._buf.shape[0]:  # grow geometrically; old contents are kept
                bigger = torch.empty(2 * self._buf.shape[0], *self._buf.shape[1:], device=self.device, dtype=torch.float32)
                bigger[: self._buf.shape[0]] = self._buf
                self._buf = bigger
            self._slot[token] = len(self._slot)
        self._buf[self._slot[token]].copy_(key_points.to(self.device, dtype=torch.float32), non_blocking=True)

    def tile(self, tokens: Sequence[int], SE3_pred: Sequence[torch.Tensor], centering_SE3: torch.Tensor
             ) -> Tuple[torch.Tensor, torch.Tensor]:
        """tokens / SE3_pred in tile order -> (tile (C, K*S) on the GPU, tokens per column (K*S,) int64 on the CPU),
        the return pair of PoseGraph.global_map_query_graph."""
        n = self._buf.shape[0]
        poses = torch.eye(4).repeat(n, 1, 1)
        sel = torch.tensor([self._slot[t] for t in tokens], dtype=torch.int32)
        for t, se3 in zip(tokens, SE3_pred):
            poses[self._slot[t]] = se3
        tile = assemble_map_tile(self._buf, poses, centering_SE3, sel)
        S = self._buf.shape[2]
        return tile, torch.tensor(list(tokens), dtype=torch.int64).repeat_interleave(S)
