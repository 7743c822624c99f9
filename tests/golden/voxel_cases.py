"""Seeded inputs of the voxel-sampler fixture (shared by make_golden_voxel.py, which runs the reference on them, and the
tests, which regenerate them): name -> (points (B,N,D), padding (B,N) bool, K, voxel_size, sample_range)."""
import torch


def _cloud(g, B, N, D, lattice=False, scale=(1.2, 1.0, 0.15)):
    pts = torch.rand(B, N, D, generator=g) * 2 - 1
    pts[..., :3] *= torch.tensor(scale)
    if lattice:  # coordinates on a 1/16 lattice: exactly equal centre distances inside voxels, equal populations
        pts[..., :3] = torch.round(pts[..., :3] * 16) / 16
    return pts


def _ragged(g, B, N):
    pad = torch.zeros(B, N, dtype=torch.bool)
    for b in range(B):
        pad[b, int(torch.randint(N // 2, N, (1,), generator=g)):] = True
    return pad


def cases():
    g = torch.Generator().manual_seed(20260929)
    out = {}
    p = _cloud(g, 2, 20000, 3)
    out["nth_element_branch"] = (p, torch.zeros(2, 20000, dtype=torch.bool), 256, 0.05, 1.0)
    out["partial_sort_branch"] = (p, torch.zeros(2, 20000, dtype=torch.bool), 16, 0.05, 1.0)
    out["ragged_features"] = (_cloud(g, 3, 6000, 5), _ragged(g, 3, 6000), 300, 0.11, 0.7)
    out["all_voxels"] = (_cloud(g, 1, 8000, 3), torch.zeros(1, 8000, dtype=torch.bool), None, 0.3, 1.0)
    out["all_voxels_fine"] = (_cloud(g, 1, 8000, 4), _ragged(g, 1, 8000), None, 0.04, 1.0)
    out["lattice_ties"] = (_cloud(g, 2, 5000, 3, lattice=True), _ragged(g, 2, 5000), 300, 0.11, 1.0)
    out["lattice_ties_large"] = (_cloud(g, 2, 12000, 5, lattice=True), torch.zeros(2, 12000, dtype=torch.bool), 64, 0.11, 1.0)
    out["fewer_than_k"] = (_cloud(g, 2, 500, 3), _ragged(g, 2, 500), 512, 0.3, 1.0)
    p = _cloud(g, 2, 3000, 3)
    p[0, :, :3] += 2.0   # frame 0 entirely outside the sample range
    out["empty_frame"] = (p, torch.zeros(2, 3000, dtype=torch.bool), 128, 0.1, 1.0)
    p = _cloud(g, 2, 4000, 3)
    p[:, 1000:2000] = p[:, :1000]   # duplicated points
    out["duplicates"] = (p, torch.zeros(2, 4000, dtype=torch.bool), 200, 0.08, 1.0)
    return out
