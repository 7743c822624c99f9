"""Registration-edge helpers of the hot path: drop-ins for the named functions of the reference's
system/modules/utils.py (information matrix, PoseTool, simvec_to_num) and the descriptor
assembly of ExtractionThread.process (system/modules/odometry.py:36-54).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def simvec_to_num(sim_vec: torch.Tensor) -> float:
    """mean of the first 30 inlier confidences (system/modules/utils.py:18)."""
    return sim_vec.flatten()[:30].mean().item()


class PoseTool(object):
    """system/modules/utils.py:30-57."""

    @classmethod
    def SE3(cls, R, t):
        if isinstance(R, np.ndarray):
            R = torch.tensor(R, dtype=torch.float32).reshape(3, 3)
        if isinstance(t, np.ndarray):
            t = torch.tensor(t, dtype=torch.float32).reshape(3, 1)
        mat = torch.eye(4)
        mat[:3, :3] = R
        mat[:3, 3:4] = t
        return mat

    @classmethod
    def Rt(cls, SE3):
        return SE3[:3, :3], SE3[:3, 3:]

    @classmethod
    def rotation_angle(cls, rot_mat) -> float:
        return torch.arccos((torch.trace(rot_mat) - 1) / 2).item()


def calculate_information_matrix_from_pcd(pointcloud_1: torch.Tensor, pointcloud_2: torch.Tensor,
                                          SE3: torch.Tensor, device="cuda") -> torch.Tensor:
    """(3,N1), (3,N2) metres, SE3 (4,4) -> (6,6) fp32 on the CPU, as the reference returns it
    (system/modules/utils.py:60-113).  Clouds may live on the CPU (ScanPack.full_pcd) or already on
    the GPU; the search and the accumulation run in libdpm_hip.so."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("calculate_information_matrix_from_pcd runs on the GPU only (no CPU fallback)")
    with torch.cuda.device(dev):
        p1 = pointcloud_1[:3].to(device=dev, dtype=torch.float32).contiguous()
        p2 = pointcloud_2[:3].to(device=dev, dtype=torch.float32).contiguous()
        se3 = SE3.to(dtype=torch.float32)
        Rt = torch.cat([se3[:3, :3].reshape(9), se3[:3, 3].reshape(3)]).to(dev).contiguous()
        return ops.information_matrix(p1, p2, Rt, 1.0).cpu()


def information_matrix_device(p1: torch.Tensor, p2: torch.Tensor, R: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """All-device variant used by the batched bench path: no host round trip."""
    Rt = torch.cat([R.reshape(9), T.reshape(3)]).contiguous()
    return ops.information_matrix(p1, p2, Rt, 1.0)


def make_descriptors(coor: torch.Tensor, fea: torch.Tensor, coor_scale: float) -> torch.Tensor:
    """ExtractionThread.process (odometry.py:47-49): cat[fea (B,128,S), xyz*coor_scale (B,3,S)] -> (B,131,S)."""
    return torch.cat([fea, coor * coor_scale], dim=1)
