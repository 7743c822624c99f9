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
