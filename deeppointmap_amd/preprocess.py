"""Scan pre-processing on the GPU, the transform chain of every shipped inference config
(configs/infer/*.yaml:21-29; reference dataloader/transforms.py):

    VoxelSample(0.3,'first') -> DistanceSample(1,60) -> OutlierFilter(10, 3.0)
        -> LowPassFilter(0.5, 16, 2.0, flux=4) -> CoordinatesNormalization(60)

    points, padding = preprocess_scan(raw_xyz)                       # voxel + distance + normalise (exact parity)
    points, padding = preprocess_scan(raw_xyz, outlier=(10, 3.0),    # the full shipped chain
                                      lowpass=(0.5, 16, 2.0, 4))
    # -> (1,3,M) normalised, (1,M) all-False: the encoder's inputs

The two filters run through pytorch3d / open3d in the reference (absent here), so their restatement is pinned by
the oracle only (oracle.outlier_filter / lowpass_filter, scipy cKDTree), not by the reference's own output.
"""
from __future__ import annotations


import torch

from . import _lib, ops

MAX_CELLS = 1 << 26  # 64 Mi voxels (256 MB of int32): 120 m x 120 m x 40 m at 0.3 m is 21 Mi


KNN_CELL = 1.0  # metres: edge of the self-kNN search grid (voxel-sampled scans have 0.3 m spacing)


def knn_self(xyz: torch.Tensor, K: int, cell: float = KNN_CELL, want=("idx", "dist2", "mean_dist")):
    """xyz (N,3) fp32 on the GPU -> dict with idx (N,K) int32 / dist2 (N,K) / mean_dist (N) of the K nearest OTHER
    points, rows ordered by (distance, index)  (pytorch3d.knn_points(p, p, K+1)[..., 1:])."""
    ops._chk(xyz, torch.float32, "xyz")
    N = xyz.shape[0]
    if N <= K:
        raise ValueError(f"need more than K={K} points, got {N}")
    lib = _lib.load()
    ws = torch.empty(lib.dpm_knn_self_workspace_bytes(N), device=xyz.device, dtype=torch.uint8)
    out = {}
    if "idx" in want:
        out["idx"] = torch.empty(N, K, device=xyz.device, dtype=torch.int32)
    if "dist2" in want:
        out["dist2"] = torch.empty(N, K, device=xyz.device, dtype=torch.float32)
    if "mean_dist" in want:
        out["mean_dist"] = torch.empty(N, device=xyz.device, dtype=torch.float32)
    _lib.check(lib.dpm_knn_self(ops._ptr(xyz), N, K, float(cell), ops._ptr(out.get("idx")), ops._ptr(out.get("dist2")),
                                ops._ptr(out.get("mean_dist")), ops._ptr(ws), ops._stream(xyz)), "dpm_knn_self")
    return out


def _stat_filter(stat, k_std, mode, xyz, idx, ratio=1.0):
    lib = _lib.load()
    N = xyz.shape[0]
    xo, io = torch.empty_like(xyz), torch.empty(N, device=xyz.device, dtype=torch.int32)
    n = torch.zeros(1, device=xyz.device, dtype=torch.int32)
    _lib.check(lib.dpm_stat_filter(ops._ptr(stat), N, float(k_std), mode, float(ratio), ops._ptr(xyz), ops._ptr(idx), ops._ptr(xo),
                                   ops._ptr(io), ops._ptr(n), ops._stream(xyz)), "dpm_stat_filter")
    m = int(n.item())  # host sync: the survivor count shapes the tensors
    return xo[:m], io[:m]


def outlier_filter(xyz: torch.Tensor, nb_neighbors: int = 10, std_ratio: float = 3.0, idx=None, ratio: float = 1.0):
    """OutlierFilter (transforms.py:230-253, pytorch3d branch): drop points whose mean distance to their
    nb_neighbors nearest neighbours exceeds mean + std_ratio * std.  xyz (N,3) metres on the GPU ->
    (kept xyz (M,3), kept original indices (M,) int32)."""
    md = knn_self(xyz, nb_neighbors, want=("mean_dist",))["mean_dist"]
    return _stat_filter(md, std_ratio, 0, xyz, idx, ratio)


def lowpass_filter(xyz: torch.Tensor, normals_radius: float = 0.5, normals_num: int = 16, filter_std: float = 2.0,
                   flux: int = 2, max_remain: int = -1, idx=None, ratio: float = 1.0):
    """LowPassFilter (transforms.py:256-289): keep points whose `flux` best normal agreements with their
    normals_num nearest neighbours sum to more than mean - filter_std * std."""
    if max_remain > 0:
        raise NotImplementedError("max_remain > 0 is not used by any shipped config and is not implemented")
    ops._chk(xyz, torch.float32, "xyz")
    lib = _lib.load()
    N = xyz.shape[0]
    ws = torch.empty(lib.dpm_knn_self_workspace_bytes(N), device=xyz.device, dtype=torch.uint8)
    normals = torch.empty(N, 3, device=xyz.device, dtype=torch.float32)
    _lib.check(lib.dpm_point_normals(ops._ptr(xyz), N, float(normals_radius), ops._ptr(normals), ops._ptr(ws),
                                     ops._stream(xyz)), "dpm_point_normals")
    nn = knn_self(xyz, normals_num, want=("idx",))["idx"]
    sim = torch.empty(N, device=xyz.device, dtype=torch.float32)
    _lib.check(lib.dpm_lowpass_similarity(ops._ptr(normals), ops._ptr(nn), N, normals_num, flux, ops._ptr(sim),
                                          ops._stream(xyz)), "dpm_lowpass_similarity")
    return _stat_filter(sim, filter_std, 1, xyz, idx, ratio)


def preprocess_scan(xyz: torch.Tensor, voxel_size: float = 0.3, min_dis: float = 1.0, max_dis: float = 60.0,
                    ratio: float = 60.0, return_index: bool = False, max_cells: int = MAX_CELLS,
                    outlier=None, lowpass=None):
    """xyz: (N,3) or (N,4) [KITTI .bin records] fp32, CPU or GPU.  Returns (points (1,3,M) fp32 on the GPU,
    padding (1,M) bool all-False[, original indices (M,) int32]).
    outlier = (nb_neighbors, std_ratio) / lowpass = (normals_radius, normals_num, filter_std, flux) switch the two
    statistical filters of the shipped chain on (they act on metres, before the division by `ratio`)."""
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
        filters = outlier is not None or lowpass is not None
        _lib.check(lib.dpm_preprocess_scan(ops._ptr(x), N, stride, float(voxel_size), float(min_dis), float(max_dis),
                                           1.0 if filters else float(ratio), int(max_cells), ops._ptr(out), ops._ptr(idx), N,
                                           ops._ptr(status), ops._ptr(ws), ops._stream(x)), "dpm_preprocess_scan")
        n_out, overflow = status.cpu().tolist()  # the one host sync: the output length shapes the tensors
    if overflow:
        raise ValueError(f"voxel grid exceeds max_cells={max_cells}; crop the scan or raise max_cells")
    if filters:
        with torch.cuda.device(dev):
            kept, kidx = out[:n_out], idx[:n_out]
            # CoordinatesNormalization (a true division, like the fused path) rides on the last filter
            if outlier is not None:
                kept, kidx = outlier_filter(kept, int(outlier[0]), float(outlier[1]), idx=kidx,
                                            ratio=1.0 if lowpass is not None else ratio)
            if lowpass is not None:
                kept, kidx = lowpass_filter(kept, float(lowpass[0]), int(lowpass[1]), float(lowpass[2]), int(lowpass[3]),
                                            idx=kidx, ratio=ratio)
            n_out, out, idx = kept.shape[0], kept, kidx
    pts = ops.to_channel_first(out[:n_out].unsqueeze(0).contiguous()) if n_out else out[:0].t().unsqueeze(0)
    pad = torch.zeros(1, n_out, dtype=torch.bool, device=dev)
    return (pts, pad, idx[:n_out]) if return_index else (pts, pad)


def preprocess_scans(scans, voxel_size: float = 0.3, min_dis: float = 1.0, max_dis: float = 60.0, ratio: float = 60.0,
                     padding_to: int = -1, max_cells: int = MAX_CELLS, streams: int = 4):
    """The head of the shipped chain (VoxelSample -> DistanceSample -> CoordinatesNormalization) for a LIST of raw scans
    ((N_i,3|4) fp32, CPU or GPU) in one go: the scans' kernels are spread over `streams` HIP streams (one voxel-grid
    workspace per stream), their output lengths come back with ONE host synchronisation, and the results are packed into
    the batch the encoder -- and the reference's multi-thread extractor, system/core.py:141-169 -- takes:
    points (B,3,M) fp32, padding (B,M) bool (True past a scan's length), lengths list.  M = the longest scan, or
    `padding_to` when positive (ToTensor(padding_to), configs/infer/*.yaml:29; scans longer than it are an error).
    Per scan the points are bit-identical to preprocess_scan()."""
    if not scans:
        raise ValueError("no scans")
    dev = torch.device("cuda", torch.cuda.current_device())
    lib = _lib.load()
    cur = torch.cuda.current_stream(dev)
    side = [torch.cuda.Stream(device=dev) for _ in range(min(streams, len(scans)))]
    work = [torch.empty(lib.dpm_preprocess_workspace_bytes(max_cells), device=dev, dtype=torch.uint8) for _ in side]
    status = torch.zeros(len(scans), 2, device=dev, dtype=torch.int32)
    outs = []
    for st in side:
        st.wait_stream(cur)
    for b, xyz in enumerate(scans):
        st = side[b % len(side)]
        with torch.cuda.stream(st):
            x = xyz.to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()
            if x.dim() != 2 or x.shape[1] < 3:
                raise ValueError("every scan must be (N,3) or (N,>=3)")
            N, stride = x.shape
            out = torch.empty(N, 3, device=dev, dtype=torch.float32)
            idx = torch.empty(N, device=dev, dtype=torch.int32)
            _lib.check(lib.dpm_preprocess_scan(ops._ptr(x), N, stride, float(voxel_size), float(min_dis), float(max_dis),
                                               float(ratio), int(max_cells), ops._ptr(out), ops._ptr(idx), N,
                                               ops._ptr(status[b]), ops._ptr(work[b % len(side)]), st.cuda_stream),
                       "dpm_preprocess_scan")
            out.record_stream(cur)  # allocated on the side stream, packed into the batch on the caller's stream below
            outs.append(out)
    for st in side:
        cur.wait_stream(st)
    st_host = status.cpu().tolist()  # the one host sync of the batch
    if any(o for _, o in st_host):
        raise ValueError(f"voxel grid exceeds max_cells={max_cells}; crop the scans or raise max_cells")
    lengths = [n for n, _ in st_host]
    M = max(lengths) if padding_to <= 0 else padding_to
    if max(lengths) > M:
        raise ValueError(f"a scan keeps {max(lengths)} points, more than padding_to={padding_to}")
    pts = torch.zeros(len(scans), 3, M, device=dev, dtype=torch.float32)
    for b, (out, n) in enumerate(zip(outs, lengths)):
        if n:
            pts[b, :, :n] = out[:n].t()
    pad = torch.arange(M, device=dev).unsqueeze(0) >= torch.tensor(lengths, device=dev).unsqueeze(1)
    return pts, pad, lengths
