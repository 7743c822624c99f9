"""Seeded synthetic RAW scan shared by make_golden.py (which feeds it to the reference) and the tests."""
import numpy as np
import torch


def raw_scan(n=120000, seed=77):
    """Metres; denser than 0.3 m voxels near the sensor, with points beyond 60 m and inside 1 m."""
    g = torch.Generator().manual_seed(seed)
    r = 0.3 + 75.0 * torch.rand(n, generator=g) ** 1.5
    th = 2 * np.pi * torch.rand(n, generator=g)
    z = -1.7 + 0.05 * r * torch.randn(n, generator=g) + 2.5 * (torch.rand(n, generator=g) < 0.2)
    return torch.stack([r * torch.cos(th), r * torch.sin(th), z], dim=1).float()
