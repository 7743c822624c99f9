"""`Sampler` / `Querier` -- the reference's string-keyed operator tables (network/encoder/utils.py:18-44,
126-147) bound to the HIP kernels.  Same construction (`Sampler('fps-t3d')`, `Querier('hybrid-t3d')`), same
keyword-only call convention (`self.sample(points=..., points_padding=..., K=...)`), same return types
(float tensors / int64 index tensors on the input's device), so reference code that uses the tables
(pointnext.py:35-36,45,49,82,91) runs unchanged.  The '-t3d' names are aliases: there is one backend.

`Sampler('voxel')` (no shipped config selects it) is the voxel-grid kernel chain of
csrc/voxel_sample.hip.  `random_start_point=True` draws its start indices from Python's `random` exactly as the reference does.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import ops


def _lengths(points_padding: torch.Tensor) -> torch.Tensor:
    return (~points_padding).sum(1).to(torch.int32).contiguous()


def _xyz(t: torch.Tensor) -> torch.Tensor:
    return t[..., :3].to(torch.float32).contiguous()


class Querier:
    def __init__(self, method: str):
        table = {"knn": self.knn_query, "ball": self.ball_query, "hybrid": self.hybrid_query,
                 "knn-t3d": self.knn_query, "ball-t3d": self.ball_query, "hybrid-t3d": self.hybrid_query}
        self.query_method = table[method.lower()]

    def __call__(self, *args, **kwargs):
        return self.query_method(**kwargs)

    @staticmethod
    def hybrid_query(radius: float, K: int, points: torch.Tensor, centers: torch.Tensor,
                     points_padding: torch.Tensor) -> torch.Tensor:
        """utils.py:76-89.  Slot 0 is the nearest point; the order of the other slots is unspecified
        (the reference returns them by ascending distance; its only consumer is a max-pool)."""
        return ops.knn_hybrid(_xyz(points), _lengths(points_padding), _xyz(centers), K, radius).long()

    @staticmethod
    def knn_query(K: int, points: torch.Tensor, centers: torch.Tensor, points_padding: torch.Tensor) -> torch.Tensor:
        """utils.py:46-54: K nearest, no radius mask."""
        return ops.knn_hybrid(_xyz(points), _lengths(points_padding), _xyz(centers), K, 1e18).long()

    @staticmethod
    def ball_query(radius: float, K: int, points: torch.Tensor, centers: torch.Tensor,
                   points_padding: torch.Tensor) -> torch.Tensor:
        """utils.py:57-73: first K indices within the radius, padded with the first."""
        return ops.ball_query(_xyz(points), _lengths(points_padding), _xyz(centers), K, radius).long()


class Sampler:
    def __init__(self, method: str):
        table = {"fps": self.fps, "fps-t3d": self.fps, "voxel": self.voxel}
        self.sample_method = table[method.lower()]

    def __call__(self, *args, **kwargs):
        return self.sample_method(**kwargs)

    @staticmethod
    def fps(points: torch.Tensor, points_padding: torch.Tensor, K: int,
            random_start_point: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """utils.py:210-285 -> (sampled points (B,K,D) with zero rows at padding, padding mask (B,K))."""
        lengths, start = _lengths(points_padding), None
        if random_start_point:
            # the reference draws random.randint(0, lengths[n] - 1) once per frame, in batch order (utils.py:248): the same
            # draws from the same generator, so a seeded run picks the same start points
            from random import randint
            start = torch.tensor([randint(0, int(n) - 1) for n in lengths.cpu()], dtype=torch.int32, device=points.device)
        idx, _, _ = ops.fps(_xyz(points), lengths, K, start=start)
        mask = idx < 0
        gathered = torch.gather(points, 1, idx.clamp(min=0).long().unsqueeze(-1).expand(-1, -1, points.shape[-1]))
        return gathered.masked_fill(mask.unsqueeze(-1), 0.0), mask

    @staticmethod
    def voxel(points: torch.Tensor, points_padding: torch.Tensor, K: int, voxel_size: float = 0.3,
              sample_range: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
        """utils.py:150-207 -> (sampled points (B,K,D): per occupied voxel the point nearest its centre, the K most
        populated voxels in torch.topk's order when there are more than K, zero rows at padding; padding mask (B,K)).
        K=None (one frame): every occupied voxel, ascending voxel id."""
        pts = points.to(torch.float32).contiguous()
        idx, _ = ops.voxel_sample(pts, points_padding, K, voxel_size, sample_range)
        mask = idx < 0
        gathered = torch.gather(points, 1, idx.clamp(min=0).long().unsqueeze(-1).expand(-1, -1, points.shape[-1]))
        return gathered.masked_fill(mask.unsqueeze(-1), 0.0), mask
