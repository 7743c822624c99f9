"""Synthetic LiDAR-like scans (BASELINE.json config 2).

frame f = a fixed base cloud seen from sensor pose P_f = translate(0.5*f m, 0, 0) o yaw(0.5 deg * f),
plus per-point jitter N(0, 0.01 m) and a row permutation (seed 1000+f), divided by 60
(the reference's CoordinatesNormalization, configs/infer/*.yaml:27).  Everything is drawn from
seeded CPU generators so the GPU box regenerates the very same bits.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

COOR_SCALE = 60.0


def base_cloud(n_points: int = 65536, seed: int = 0) -> torch.Tensor:
    """(n_points, 3) metres: r = 1 + 59*sqrt(u1), theta = 2*pi*u2, z ~ N(0, 1 m)."""
    g = torch.Generator().manual_seed(seed)
    u1 = torch.rand(n_points, generator=g, dtype=torch.float64)
    u2 = torch.rand(n_points, generator=g, dtype=torch.float64)
    z = torch.randn(n_points, generator=g, dtype=torch.float64)
    r = 1.0 + 59.0 * torch.sqrt(u1)
    th = 2.0 * math.pi * u2
    return torch.stack([r * torch.cos(th), r * torch.sin(th), z], dim=1)


def sensor_pose(f: int) -> torch.Tensor:
    """4x4 float64 pose of the sensor at frame f in the world frame."""
    a = math.radians(0.5 * f)
    P = torch.eye(4, dtype=torch.float64)
    P[0, 0], P[0, 1], P[1, 0], P[1, 1] = math.cos(a), -math.sin(a), math.sin(a), math.cos(a)
    P[0, 3] = 0.5 * f
    return P


def frame(f: int, n_points: int = 65536, base: torch.Tensor = None) -> torch.Tensor:
    """(3, n_points) float32, normalised coordinates (metres / 60) of frame f."""
    if base is None:
        base = base_cloud(n_points)
    P = sensor_pose(f)
    R, t = P[:3, :3], P[:3, 3]
    pts = (base - t) @ R  # == (R^T (x - t))^T : world -> sensor
    g = torch.Generator().manual_seed(1000 + f)
    pts = pts + 0.01 * torch.randn(pts.shape, generator=g, dtype=torch.float64)
    perm = torch.randperm(n_points, generator=g)
    pts = pts[perm]
    return (pts / COOR_SCALE).to(torch.float32).t().contiguous()


def frames(n_frames: int, n_points: int = 65536, start: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """(points (F,3,N) f32, padding (F,N) bool all-False) -- the encoder's input contract."""
    base = base_cloud(n_points)
    pts = torch.stack([frame(start + f, n_points, base) for f in range(n_frames)], dim=0)
    return pts, torch.zeros(n_frames, n_points, dtype=torch.bool)


def relative_pose(f_src: int, f_dst: int) -> torch.Tensor:
    """Ground-truth 4x4 taking points of frame f_src into frame f_dst (metres)."""
    return torch.linalg.inv(sensor_pose(f_dst)) @ sensor_pose(f_src)
