"""Scan pre-processing on the GPU: VoxelSample('first') -> DistanceSample -> CoordinatesNormalization, the head of
every shipped inference transform chain (reference dataloader/transforms.py:322-356,387-407;
configs/infer/*.yaml:21-27).  OutlierFilter / LowPassFilter (pytorch3d / open3d based) are not part of this step.

    points, padding = preprocess_scan(raw_xyz)        # -> (1,3,M) normalised, (1,M) all-False: the encoder's inputs
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib, ops

MAX_CELLS = 1 << 26  # 64 Mi voxels (256 MB of int32): 120 m x 120 m x 40 m at 0.3 m is 21 Mi


def preprocess_scan(xyz: torch.Tensor, voxel_size: float = 0.3, min_dis: float = 1.0, max_dis: float = 60.0,
                    ratio: float = 60.0, return_index: bool = False, max_cells: int = MAX_CELLS):
    """xyz: (N,3) or (N,4) [KITTI .bin records] fp32, CPU or GPU.  Returns (points (1,3,M) fp32 on the GPU,
    padding (1,M) bool all-False[, original indices (M,) int32])."""
    dev = xyz.device if xyz.is_cuda else torch.device("cuda", torch.cuda.current_device())
    x = xyz.to(device=dev, dtype=torch.float32).contiguous()
    if x.dim() != 2 or x.shape[1] < 3:
        raise ValueError("xyz must be (N,3) or (N,>=3)")
    N, stride = x.shape
    lib = _lib.load()
    with torch.cuda.device(dev):
        ws = torch.empty(lib.dpm_preprocess_workspace_bytes(max_cells), device=dev, dtype=torch.uint8)
        out = torch.empty(N, 3, device=dev, dtype=torch.float32)
        idx = torch.empty(N, device=dev, dtype=torch.int32)
        status = torch.zeros(2, device=dev, dtype=torch.int32)
        _lib.check(lib.dpm_preprocess_scan(ops._ptr(x), N, stride, float(voxel_size), float(min_dis), float(max_dis),
                                           float(ratio), int(max_cells), ops._ptr(out), ops._ptr(idx), N,
                                           ops._ptr(status), ops._ptr(ws), ops._stream(x)), "dpm_preprocess_scan")
        n_out, overflow = status.cpu().tolist()  # the one host sync: the output length shapes the tensors
    if overflow:
        raise ValueError(f"voxel grid exceeds max_cells={max_cells}; crop the scan or raise max_cells")
    pts = ops.to_channel_first(out[:n_out].unsqueeze(0).contiguous()) if n_out else out[:0].t().unsqueeze(0)
    pad = torch.zeros(1, n_out, dtype=torch.bool, device=dev)
    return (pts, pad, idx[:n_out]) if return_index else (pts, pad)
