"""Scan pre-processing: oracle vs the reference's own transform classes (CPU); HIP kernels vs
the same fixtures (GPU)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, T, load_golden
from oracle import dpm_oracle as O


def _raw_scan():
    sys.path.insert(0, GOLDEN)
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_mg_raw", os.path.join(GOLDEN, "raw_scan.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.raw_scan()
    finally:
        sys.path.pop(0)


def _cases():
    g = load_golden("preprocess.npz")
    return g, {"raw120k": _raw_scan(), "kitti0_m": T(g["kitti0_m.in"])}


def test_oracle_preprocess_vs_reference_transforms():
    g, cases = _cases()
    for name, xyz in cases.items():
        out, idx = O.preprocess_scan(xyz)
        assert np.array_equal(out.numpy(), g[name + ".out"]), name
    assert g["raw120k.out"].shape[0] > 20000  # the case really thins a dense scan and crops both ends


@pytest.mark.gpu
def test_hip_preprocess_vs_reference_transforms():
    from deeppointmap_amd.preprocess import preprocess_scan
    g, cases = _cases()
    for name, xyz in cases.items():
        pts, pad, idx = preprocess_scan(xyz, return_index=True)
        want = g[name + ".out"]
        assert tuple(pts.shape) == (1, 3, want.shape[0]) and not bool(pad.any()), name
        assert np.array_equal(pts[0].t().cpu().numpy(), want), name  # same points, same order, same bits
        assert torch.equal(xyz[idx.cpu().long()] / 60.0, pts[0].t().cpu())
    # KITTI .bin layout (N,4) goes through the same kernels with stride 4
    xyz = cases["raw120k"]
    p4, _ = preprocess_scan(torch.cat([xyz, torch.ones(xyz.shape[0], 1)], dim=1))
    assert np.array_equal(p4[0].t().cpu().numpy(), g["raw120k.out"])
    with pytest.raises(ValueError, match="max_cells"):
        preprocess_scan(xyz, max_cells=4096)


@pytest.mark.gpu
def test_hip_distance_crop_on_the_radius():
    """DistanceSample keeps min <= torch.norm(xyz, dim=1) <= max (transforms.py:394-395).  360 000 points within six ulps of the
    60 m radius: the device crop must keep exactly the points the reference's own expression keeps -- torch's CPU norm is a
    float fma chain over x, y, z, and a sum-of-squares in another order is off by an ulp for one point in ten."""
    from deeppointmap_amd.preprocess import preprocess_scan
    from oracle import dpm_oracle as O
    g = torch.Generator().manual_seed(3)
    for trial in range(6):
        n = 60000
        d = torch.randn(n, 3, generator=g)
        d = d / d.norm(dim=1, keepdim=True)
        x = d * (60.0 + torch.randint(-6, 7, (n, 1), generator=g).float() * 3.8e-6) + (torch.rand(n, 3, generator=g) - 0.5) * 1e-5
        dis = torch.norm(x, p=2, dim=1)
        assert 0.2 < float(((1.0 <= dis) & (dis <= 60.0)).float().mean()) < 0.8          # the radius cuts through the cloud
        wp, wi = O.preprocess_scan(x, 0.4, 1.0, 60.0)
        assert bool(((1.0 <= dis[wi.long()]) & (dis[wi.long()] <= 60.0)).all())            # (the oracle is that expression)
        pts, pad, idx = preprocess_scan(x, 0.4, 1.0, 60.0, return_index=True)
        assert pts.shape[2] == wp.shape[0] and np.array_equal(idx.cpu().numpy(), wi.numpy().astype(np.int32)), trial


# ------------------------------------------------------------------------------------------------
# OutlierFilter / LowPassFilter (transforms.py:230-289): pytorch3d / open3d based in the reference, so the checker
# is the oracle's restatement (scipy cKDTree + numpy eigh); parity with the reference itself is unpinned.
# ------------------------------------------------------------------------------------------------
def _plane_with_outliers():
    gx, gy = torch.meshgrid(torch.arange(40.0), torch.arange(40.0), indexing="ij")
    plane = torch.stack([gx.flatten() * 0.3, gy.flatten() * 0.3, torch.zeros(1600)], dim=1)
    gen = torch.Generator().manual_seed(3)
    plane[:, :2] += 0.01 * torch.randn(1600, 2, generator=gen)
    lone = torch.tensor([[6.0, 6.0, 5.0], [3.0, 9.0, -4.0]])
    return torch.cat([plane, lone])


def test_oracle_filters_on_hand_cases():
    xyz = _plane_with_outliers()
    idx, d2 = O.knn_self(xyz, 4)
    brute = ((xyz[:, None] - xyz[None]) ** 2).sum(-1)
    brute.fill_diagonal_(float("inf"))
    want, _ = torch.sort(brute, dim=1)
    torch.testing.assert_close(d2, want[:, :4], rtol=1e-5, atol=1e-7)
    keep = O.outlier_filter(xyz, 10, 3.0)
    assert bool(keep[:1600].all()) and not bool(keep[1600:].any())       # exactly the two lone points go
    n = O.point_normals(xyz[:1600], 0.5)
    assert float(n[:, 2].abs().min()) > 0.999                           # a plane: every normal is +-z
    wall = xyz[:1600][:, [0, 2, 1]] + torch.tensor([0.0, 30.0, 0.0])    # the same patch standing up, far away
    gen = torch.Generator().manual_seed(4)
    fuzz = torch.tensor([20.0, 20.0, 0.0]) + 0.6 * torch.randn(150, 3, generator=gen)   # a bush: no coherent normal
    both = torch.cat([xyz[:1600], wall, fuzz])
    keep, sim = O.lowpass_filter(both, 0.5, 16, 2.0, 4)
    assert float(sim[:3200].min()) > 3.99 and bool(keep[:3200].all())  # smooth surfaces pass untouched
    assert float(sim[3200:].mean()) < 3.9 and int((~keep[3200:]).sum()) > 30  # the cut falls inside the bush


def _outlier_cloud():
    base, _ = O.preprocess_scan(_raw_scan(), ratio=1.0)                # metres, after VoxelSample + DistanceSample
    gen = torch.Generator().manual_seed(5)
    return torch.cat([base, 40 * torch.rand(30, 3, generator=gen) + torch.tensor([0.0, 0.0, 8.0])]).contiguous()


def test_oracle_outlier_filter_vs_reference_code():
    """oracle.outlier_filter against the REFERENCE's OutlierFilter (pytorch3d branch, its knn_points call answered by
    exhaustive search: tests/golden/make_golden_infomat.py): the same survivors"""
    g = load_golden("outlier_filter.npz")
    xyz = _outlier_cloud()
    assert xyz.shape[0] == int(g["n"])
    keep = O.outlier_filter(xyz, 10, 3.0)
    assert 0 < int((~keep).sum()) < xyz.shape[0] // 20
    assert np.array_equal(keep.numpy(), g["keep"])


@pytest.mark.gpu
def test_hip_outlier_filter_vs_reference_code():
    from deeppointmap_amd import preprocess as P
    g = load_golden("outlier_filter.npz")
    xyz = _outlier_cloud()
    kx, ki = P.outlier_filter(xyz.to("cuda:0"), 10, 3.0)
    assert torch.equal(ki.cpu().long(), torch.nonzero(T(g["keep"])).flatten())
    assert torch.equal(kx.cpu(), xyz[T(g["keep"])])


@pytest.mark.gpu
def test_hip_filters_vs_oracle():
    from deeppointmap_amd import preprocess as P
    dev = torch.device("cuda:0")
    base, _ = O.preprocess_scan(_raw_scan(), ratio=1.0)                # metres, after VoxelSample + DistanceSample
    gen = torch.Generator().manual_seed(5)
    xyz = torch.cat([base, 40 * torch.rand(30, 3, generator=gen) + torch.tensor([0.0, 0.0, 8.0])]).contiguous()
    g = xyz.to(dev)
    N = xyz.shape[0]
    # exact K nearest other points: distances to the bit, index rows equal up to exact distance ties
    for K in (10, 16):
        widx, wd2 = O.knn_self(xyz, K)
        out = P.knn_self(g, K)
        assert torch.equal(out["dist2"].cpu(), wd2), K
        same = (out["idx"].cpu().long() == widx).all(1)
        assert float(same.float().mean()) > 0.999
        torch.testing.assert_close(out["mean_dist"].cpu(), torch.sqrt(wd2).mean(1), rtol=2e-6, atol=0)
    # OutlierFilter: same survivors (a point exactly on the cut may flip: none expected)
    keep = O.outlier_filter(xyz, 10, 3.0)
    kx, ki = P.outlier_filter(g, 10, 3.0)
    assert 0 < int((~keep).sum()) < N // 20
    assert torch.equal(ki.cpu().long(), torch.nonzero(keep).flatten())
    assert torch.equal(kx.cpu(), xyz[keep])
    # normals: same direction up to sign wherever the smallest eigenvalue is well separated
    import deeppointmap_amd._lib as L
    from deeppointmap_amd import ops
    lib = L.load()
    ws = torch.empty(lib.dpm_knn_self_workspace_bytes(N), device=dev, dtype=torch.uint8)
    nrm = torch.empty(N, 3, device=dev)
    L.check(lib.dpm_point_normals(ops._ptr(g), N, 0.5, ops._ptr(nrm), ops._ptr(ws), ops._stream(g)), "dpm_point_normals")
    wn = O.point_normals(xyz, 0.5)
    agree = (nrm.cpu() * wn).sum(1).abs()
    assert float((agree > 1 - 1e-4).float().mean()) > 0.97
    torch.testing.assert_close(nrm.cpu().norm(dim=1), torch.ones(N), rtol=0, atol=1e-5)
    # LowPassFilter on the SAME normals: the similarity statistic and the survivors agree
    keep, sim = O.lowpass_filter(xyz, 0.5, 16, 2.0, 4, normals=nrm.cpu())
    nn = P.knn_self(g, 16, want=("idx",))["idx"]
    gsim = torch.empty(N, device=dev)
    L.check(lib.dpm_lowpass_similarity(ops._ptr(nrm), ops._ptr(nn), N, 16, 4, ops._ptr(gsim), ops._stream(g)), "sim")
    close = (gsim.cpu() - sim).abs() < 1e-5
    assert float(close.float().mean()) > 0.999                          # rows with an exact kNN tie may differ
    kx, ki = P.lowpass_filter(g, 0.5, 16, 2.0, 4)
    got = torch.zeros(N, dtype=torch.bool)
    got[ki.cpu().long()] = True
    assert int((got != O.lowpass_filter(xyz, 0.5, 16, 2.0, 4)[0]).sum()) <= N // 200   # own normals on both sides
    # the full shipped chain through the one entry point
    pts, pad, idx = P.preprocess_scan(_raw_scan(), outlier=(10, 3.0), lowpass=(0.5, 16, 2.0, 4), return_index=True)
    M = pts.shape[2]
    assert 0.8 * base.shape[0] < M < base.shape[0] and not bool(pad.any())
    assert torch.equal(pts[0].t().cpu(), _raw_scan()[idx.cpu().long()] / 60.0)
