"""Thin torch-tensor wrappers over the C ABI (include/dpm_hip.h).

torch provides device memory and the stream; every computation happens inside libdpm_hip.so.
Tensors must live on a ROCm device ("cuda" in torch terms); layouts are point-major fp32,
indices int32.  Nothing here falls back to torch math.
"""
from __future__ import annotations

from typing import Optional, Tuple

import ctypes
import threading
import weakref

import torch

from . import _lib, knobs

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.DpmError(f"{name}: expected a tensor on the GPU, got {t.device} (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def prepare_points(points_cf: torch.Tensor, padding: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(B,C,N) f32 + (B,N) bool -> xyz (B,N,3), lengths (B,) int32."""
    _chk(points_cf, torch.float32, "points")
    pad = _chk(padding.view(torch.uint8) if padding.dtype == torch.bool else padding, torch.uint8, "padding")  # same bytes
    B, C, N = points_cf.shape
    xyz = torch.empty(B, N, 3, device=points_cf.device, dtype=torch.float32)
    lengths = torch.empty(B, device=points_cf.device, dtype=torch.int32)
    lib = _lib.load()
    _lib.check(lib.dpm_prepare_points(_ptr(points_cf), _ptr(pad), B, C, N, _ptr(xyz), _ptr(lengths),
                                      _stream(xyz)), "dpm_prepare_points")
    return xyz, lengths


def emit_descriptors(xyz: torch.Tensor, fea: torch.Tensor, lengths: torch.Tensor, coor_scale: float = 0.0,
                     spare_frames: int = 0):
    """Point-major last level -> the encoder's return triple (coor (B,3,S), feat (B,C,S), padding (B,S) bool) and, with
    coor_scale > 0, the unified descriptor (B,C+3,S) = [feat ; coor * coor_scale] (odometry.py:47-49); one launch."""
    _chk(xyz, torch.float32, "xyz"), _chk(fea, torch.float32, "fea"), _chk(lengths, torch.int32, "lengths")
    B, S, C = fea.shape
    dev = xyz.device
    coor = torch.empty(B, 3, S, device=dev, dtype=torch.float32)
    feat = torch.empty(B, C, S, device=dev, dtype=torch.float32)
    padding = torch.empty(B, S, device=dev, dtype=torch.bool)
    # spare_frames: extra descriptor slots behind the B written ones (a neighbour rank's hand-over frame goes there)
    desc = torch.empty(B + spare_frames, C + 3, S, device=dev, dtype=torch.float32) if coor_scale > 0 else None
    _lib.check(_lib.load().dpm_emit_descriptors(_ptr(xyz), _ptr(fea), _ptr(lengths), B, S, C, float(coor_scale), _ptr(coor),
                                                _ptr(feat), _ptr(padding), _ptr(desc), _stream(xyz)), "dpm_emit_descriptors")
    return coor, feat, padding, desc


def nested_levels(xyz0: torch.Tensor, len0: torch.Tensor, npoints):
    """Levels below the first as prefixes of its picks: -> [(idx (B,K) int32, xyz (B,K,3), lengths (B,)) per K in npoints],
    views of three packed buffers written by one launch."""
    _chk(xyz0, torch.float32, "xyz0"), _chk(len0, torch.int32, "len0")
    B, K0, _ = xyz0.shape
    npoints = [int(k) for k in npoints]
    tot = B * sum(npoints)
    dev = xyz0.device
    xyz = torch.empty(tot, 3, device=dev, dtype=torch.float32)
    idx = torch.empty(tot, device=dev, dtype=torch.int32)
    lens = torch.empty(len(npoints), B, device=dev, dtype=torch.int32)
    arr = (ctypes.c_int32 * len(npoints))(*npoints)
    _lib.check(_lib.load().dpm_nested_levels(_ptr(xyz0), _ptr(len0), B, K0, len(npoints), ctypes.cast(arr, ctypes.c_void_p),
                                             _ptr(xyz), _ptr(idx), _ptr(lens), _stream(xyz0)), "dpm_nested_levels")
    out, off = [], 0
    for i, K in enumerate(npoints):
        out.append((idx[off:off + B * K].view(B, K), xyz[off:off + B * K].view(B, K, 3), lens[i]))
        off += B * K
    return out


def gather_frames(src: torch.Tensor, index: torch.Tensor, rows: int, cols: int, ld: int = None, frame_stride: int = None,
                  offset: int = 0) -> torch.Tensor:
    """out (n, rows, cols) = rows [0, rows) x columns [offset, offset + cols) of the frames index[p] of `src`
    (frames `frame_stride` floats apart, rows `ld` floats apart; defaults: packed)."""
    _chk(index, torch.int32, "index")
    if ld is None:
        _chk(src, torch.float32, "src")
    elif not src.is_cuda or src.dtype != torch.float32 or src.stride(-1) != 1:   # an explicit row stride: a row view
        raise ValueError("src: expected an fp32 GPU tensor with unit column stride")
    ld = cols if ld is None else ld
    frame_stride = rows * ld if frame_stride is None else frame_stride
    n = index.numel()
    out = torch.empty(n, rows, cols, device=src.device, dtype=torch.float32)
    base = src.data_ptr() + 4 * offset
    _lib.check(_lib.load().dpm_gather_frames(ctypes.c_void_p(base), frame_stride, rows, ld, cols, _ptr(index), n, _ptr(out),
                                             _stream(src)), "dpm_gather_frames")
    return out


def to_channel_first(x: torch.Tensor, row_multiple: int = 1) -> torch.Tensor:
    """(B,R,C) -> (B,C,R).  row_multiple > 1: the output rows are padded to a multiple of that many floats and the result
    is the (B,C,R) view of the padded buffer (row stride > R)."""
    _chk(x, torch.float32, "x")
    B, R, C = x.shape
    ldo = -(-R // row_multiple) * row_multiple
    out = torch.empty(B, C, ldo, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().dpm_to_channel_first_ld(_ptr(x), B, R, C, _ptr(out), ldo, _stream(x)), "dpm_to_channel_first")
    return out[:, :, :R] if ldo != R else out


def fps(xyz: torch.Tensor, lengths: torch.Tensor, K: int, algo: int = 0, start: Optional[torch.Tensor] = None):
    """xyz (B,N,3), lengths (B,) -> idx (B,K) int32 (-1 = padding), new_xyz (B,K,3), new_lengths (B,).
    start (B,) int32: first pick of every frame (`random_start_point`); None = point 0."""
    _chk(xyz, torch.float32, "xyz")
    _chk(lengths, torch.int32, "lengths")
    B, N, _ = xyz.shape
    lib = _lib.load()
    if start is not None:
        _chk(start, torch.int32, "start")
        if tuple(start.shape) != (B,):
            raise ValueError(f"start must have shape ({B},)")
        idx = torch.empty(B, K, device=xyz.device, dtype=torch.int32)
        new_xyz = torch.empty(B, K, 3, device=xyz.device, dtype=torch.float32)
        new_len = torch.empty(B, device=xyz.device, dtype=torch.int32)
        ws = torch.empty(lib.dpm_fps_workspace_bytes(B, N, K), device=xyz.device, dtype=torch.uint8)
        _lib.check(lib.dpm_fps_start(_ptr(xyz), _ptr(lengths), _ptr(start), B, N, K, _ptr(idx), _ptr(new_xyz), _ptr(new_len),
                                     _ptr(ws), _stream(xyz)), "dpm_fps_start")
        return idx, new_xyz, new_len
    if algo == 0 and N > 16384 and knobs.FPS_ALGO is not None:  # A/B runs (scripts/); never set from the environment here
        algo = int(knobs.FPS_ALGO)
    idx = torch.empty(B, K, device=xyz.device, dtype=torch.int32)
    new_xyz = torch.empty(B, K, 3, device=xyz.device, dtype=torch.float32)
    new_len = torch.empty(B, device=xyz.device, dtype=torch.int32)
    ws = torch.empty(lib.dpm_fps_workspace_bytes(B, N, K), device=xyz.device, dtype=torch.uint8)
    _lib.check(lib.dpm_fps_ex(_ptr(xyz), _ptr(lengths), B, N, K, _ptr(idx), _ptr(new_xyz), _ptr(new_len),
                              _ptr(ws), algo, _stream(xyz)), "dpm_fps")
    return idx, new_xyz, new_len


GRID_MIN_N = 1024  # dpm_knn_hybrid searches a grid from this many points per frame on


def knn_grid(points: torch.Tensor, lengths: torch.Tensor, radius: float) -> torch.Tensor:
    """The point-only half of knn_hybrid (N >= GRID_MIN_N): sorts every frame into the search grid of `radius`.
    Returns the workspace to hand to ONE knn_hybrid(..., grid=...) call with the same points, lengths and radius."""
    _chk(points, torch.float32, "points"), _chk(lengths, torch.int32, "lengths")
    B, N, _ = points.shape
    lib = _lib.load()
    ws = torch.empty(lib.dpm_knn_workspace_bytes(B, N), device=points.device, dtype=torch.uint8)
    _lib.check(lib.dpm_knn_build_grid(_ptr(points), _ptr(lengths), B, N, float(radius), _ptr(ws), _stream(points)),
               "dpm_knn_build_grid")
    return ws


def knn_hybrid(points: torch.Tensor, lengths: torch.Tensor, centers: torch.Tensor, K: int,
               radius: float, brute: bool = False, reuse_idx: Optional[torch.Tensor] = None,
               center_src: Optional[torch.Tensor] = None, grid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """points (B,N,3), centers (B,S,3) -> idx (B,S,K) int32.  brute=True forces the all-pairs scan.
    reuse_idx (B,N,K) + center_src (B,S): centres that are points of the frame (center_src >= 0) copy their row of
    the already computed self-query (same radius, same K); only padded centres are searched.
    grid: knn_grid(points, lengths, radius) built ahead of time -- only the search runs."""
    _chk(points, torch.float32, "points")
    _chk(centers, torch.float32, "centers")
    _chk(lengths, torch.int32, "lengths")
    B, N, _ = points.shape
    S = centers.shape[1]
    idx = torch.empty(B, S, K, device=points.device, dtype=torch.int32)
    lib = _lib.load()
    if grid is not None:
        if brute or reuse_idx is not None or grid.numel() != lib.dpm_knn_workspace_bytes(B, N):
            raise ValueError("grid= goes with a plain grid search of the (B, N) it was built for")
        _lib.check(lib.dpm_knn_hybrid_prebuilt(_ptr(points), _ptr(lengths), _ptr(centers), B, N, S, K, float(radius),
                                               _ptr(idx), _ptr(grid), _stream(points)), "dpm_knn_hybrid_prebuilt")
        return idx
    nbytes = 0 if brute else lib.dpm_knn_workspace_bytes(B, N)
    ws = torch.empty(nbytes, device=points.device, dtype=torch.uint8) if nbytes else None
    if reuse_idx is not None:
        _chk(reuse_idx, torch.int32, "reuse_idx"), _chk(center_src, torch.int32, "center_src")
        if tuple(reuse_idx.shape) != (B, N, K) or tuple(center_src.shape) != (B, S):
            raise ValueError("reuse_idx must be (B,N,K) and center_src (B,S)")
    _lib.check(lib.dpm_knn_hybrid_reuse(_ptr(points), _ptr(lengths), _ptr(centers), B, N, S, K, float(radius),
                                        _ptr(idx), _ptr(ws), _ptr(reuse_idx), _ptr(center_src), _stream(points)),
               "dpm_knn_hybrid")
    return idx


def ball_query(points: torch.Tensor, lengths: torch.Tensor, centers: torch.Tensor, K: int, radius: float) -> torch.Tensor:
    """points (B,N,3), centers (B,S,3) -> idx (B,S,K) int32: first K indices within the radius, padded with the first."""
    _chk(points, torch.float32, "points"), _chk(centers, torch.float32, "centers"), _chk(lengths, torch.int32, "lengths")
    B, N, _ = points.shape
    S = centers.shape[1]
    idx = torch.empty(B, S, K, device=points.device, dtype=torch.int32)
    _lib.check(_lib.load().dpm_ball_query(_ptr(points), _ptr(lengths), _ptr(centers), B, N, S, K, float(radius),
                                          _ptr(idx), _stream(points)), "dpm_ball_query")
    return idx


VOXEL_SAMPLER_MAX_CELLS = 1 << 28   # 3 GB of grid per frame; finer grids than that are refused
VOXEL_SAMPLER_MAX_WORKSPACE = 64 << 30   # ... and so is a batch whose grids together exceed 64 GiB


def voxel_sample(points: torch.Tensor, padding: torch.Tensor, K: Optional[int], voxel_size: float = 0.3,
                 sample_range: float = 1.0):
    """Sampler.voxel (utils.py:150-207): points (B,N,D) float32, padding (B,N) bool -> (sel (B,cap) int32 original
    indices in the reference's output order, -1 = padding; n_unique (B,) int32 occupied voxels).  cap = K, or the number
    of occupied voxels for K=None (B must be 1 then, as in the reference).  One host read of the grid sizes (the
    reference walks every frame on the host), a second one for K=None."""
    _chk(points, torch.float32, "points")
    if padding.dtype != torch.bool or tuple(padding.shape) != tuple(points.shape[:2]):
        raise ValueError("points_padding must be a (B,N) bool tensor")
    B, N, D = points.shape
    if K is None and B != 1:
        raise ValueError("K=None takes one frame (utils.py:200)")
    lib, dev, st = _lib.load(), points.device, _stream(points)
    pad = padding.contiguous().view(torch.uint8)
    hdr = torch.empty(B, 8, device=dev, dtype=torch.float32)
    _lib.check(lib.dpm_voxel_sampler_bounds(_ptr(points), _ptr(pad), B, N, D, float(voxel_size), float(sample_range),
                                            _ptr(hdr), st), "dpm_voxel_sampler_bounds")
    dims = hdr[:, 3:6].double().cpu()
    if not bool(torch.isfinite(dims).all()):
        raise ValueError("voxel sampler: the frame's bounding box is not finite")
    cells = int(dims.prod(1).max().item())
    if cells > VOXEL_SAMPLER_MAX_CELLS:
        raise ValueError(f"voxel sampler: {cells} grid cells per frame exceed {VOXEL_SAMPLER_MAX_CELLS}")
    ws_bytes = lib.dpm_voxel_sampler_workspace_bytes(B, N, cells)
    if ws_bytes > VOXEL_SAMPLER_MAX_WORKSPACE:
        raise ValueError(f"voxel sampler: {B} frames x {cells} grid cells need {ws_bytes >> 20} MiB of grid "
                         f"(limit {VOXEL_SAMPLER_MAX_WORKSPACE >> 20} MiB): sample fewer frames per call or use a coarser grid")
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    cap = N if K is None else int(K)
    sel = torch.empty(B, cap, device=dev, dtype=torch.int32)
    n_unique = torch.empty(B, device=dev, dtype=torch.int32)
    _lib.check(lib.dpm_voxel_sampler_select(_ptr(points), _ptr(pad), B, N, D, float(voxel_size), float(sample_range),
                                            _ptr(hdr), cells, -1 if K is None else int(K), _ptr(sel), cap,
                                            _ptr(n_unique), _ptr(ws), st), "dpm_voxel_sampler_select")
    if K is None:
        sel = sel[:, :int(n_unique[0].item())]
    return sel, n_unique


PROJECTED_COUT = (32, 64, 128, 256, 512)

# Tensors derived from weights alone (a packed copy of weight columns, the product of two weight matrices) are
# made once per weight version, not once per call.  An entry belongs to the very tensor OBJECTS it was made from
# (weak references: a freed weight whose address is reused cannot alias it) at their storage address and in-place
# version, so load_state_dict / .to() / optimiser steps invalidate it.  Callers pass the parameter tensors
# themselves, not views made per call.
_DERIVED: dict = {}
_DERIVED_LOCK = threading.Lock()   # the reference drives one Encoder / Decoder from several threads (core.py:54-57)
_RETIRED: list = []                # replaced values stay alive until the device has been synchronised once more
_GENERATION = [0]                  # bumped by invalidate_derived(): captured graphs made before it are stale (decoder.py)


def invalidate_derived() -> None:
    """Forget every tensor derived from weights.  load_state_dict() calls it (ParamTree); call it yourself after editing
    weights through `.data` (which does not bump the version counter the cache keys on)."""
    with _DERIVED_LOCK:
        _RETIRED.extend(v[2] for v in _DERIVED.values())
        _DERIVED.clear()
        _GENERATION[0] += 1


def derived_generation() -> int:
    return _GENERATION[0]


def _derived(tag: str, sources, make):
    key = (tag,) + tuple(id(t) for t in sources)
    stamp = tuple((t.data_ptr(), t._version) for t in sources)
    dev = sources[0].device
    cur = torch.cuda.current_stream(dev)
    if torch.cuda.is_current_stream_capturing():
        # Inside a capture the cache is read-only: a tensor MADE here would live in the graph's private pool, hold no data
        # until the first replay and be published to eager callers with a captured event; and a hit must not make the
        # capture wait on an event recorded outside it -- it has to be complete already (Decoder captures a shape only after
        # eager calls of the same shape, which made every derived tensor of the path long before).
        with _DERIVED_LOCK:
            hit = _DERIVED.get(key)
            if hit is not None and hit[1] == stamp and all(r() is t for r, t in zip(hit[0], sources)) and hit[3].query():
                return hit[2]
        raise RuntimeError(f"ops._derived({tag!r}) under stream capture: make weight-derived tensors before capturing")
    with _DERIVED_LOCK:
        hit = _DERIVED.get(key)
        if hit is not None and hit[1] == stamp and all(r() is t for r, t in zip(hit[0], sources)):
            cur.wait_event(hit[3])       # made on another stream, possibly moments ago
            for v in (hit[2] if isinstance(hit[2], (tuple, list)) else (hit[2],)):
                v.record_stream(cur)     # ... and read on this one: the allocator must not recycle it under the reader
            return hit[2]
        if len(_RETIRED) > 64:           # rare: bounded by a device sync, after which nothing can still read them
            torch.cuda.synchronize(dev)
            _RETIRED.clear()
        if hit is not None:
            _RETIRED.append(hit[2])
        if len(_DERIVED) >= 512:
            # entries of weights that no longer exist go first (models re-created, ad-hoc weights); nothing can still read them
            # through a captured graph, whose owner holds its weights.  If live entries have to go as well, captured graphs
            # that replay kernels reading them are stale: the generation they were stamped with ends here (decoder.py).
            dead = [k for k, v in _DERIVED.items() if any(r() is None for r in v[0])]
            for k in dead:
                _RETIRED.append(_DERIVED.pop(k)[2])
            if len(_DERIVED) >= 512:
                _RETIRED.extend(v[2] for v in _DERIVED.values())
                _DERIVED.clear()
                _GENERATION[0] += 1
        value = make()
        _DERIVED[key] = (tuple(weakref.ref(t) for t in sources), stamp, value, cur.record_event())
        return value


def _centre_rows(M: torch.Tensor) -> torch.Tensor:
    """(Cout, n) -> the same with every column's mean over the Cout rows removed (fp64 arithmetic, fp32 result, contiguous)"""
    Md = M.double()
    return (Md - Md.mean(0, keepdim=True)).float().contiguous()


def _gamma_sign(gamma: torch.Tensor) -> torch.Tensor:
    """(Cout, 1): +1 where gamma >= 0, -1 where gamma < 0"""
    return torch.where(gamma < 0, -torch.ones_like(gamma), torch.ones_like(gamma)).reshape(-1, 1)


def _centred_layer(W2: torch.Tensor, bias: torch.Tensor, gamma: torch.Tensor, Cin: int):
    """diag(sign gamma) (I - 11^T / Cout) applied to a grouping layer (csrc/group_mlp.hip, CENTRED):
    -> (W_f (Cout,Cin), W_r (Cout,3), bias (Cout)): zero mean over Cout, then channel c times sign(gamma_c)"""
    sg = _gamma_sign(gamma)
    Wc = _centre_rows(W2) * sg
    return Wc[:, :Cin].contiguous(), Wc[:, Cin:].contiguous(), (_centre_rows(bias.reshape(-1, 1)) * sg).reshape(-1).contiguous()


def group_mlp_max(xyz, fea, centers, idx, W, bias, gamma, beta, radius: float, generic: bool = False,
                  fused: bool = False) -> torch.Tensor:
    """xyz (B,N,3), fea (B,N,Cin), centers (B,S,3), idx (B,S,K), W (Cout,Cin+3[,1,1]) -> (B,S,Cout).
    Default: project before gather (P = fea W_f^T + b once per point with the MFMA GEMM, then gather + relative
    coordinates + LayerNorm + max).  fused=True: the one-kernel gather-GEMM path; generic=True: the plain-VALU
    kernel (cross-check paths, and the fallback for layer widths the projected path does not cover)."""
    for n, t in (("xyz", xyz), ("fea", fea), ("centers", centers), ("W", W), ("bias", bias),
                 ("gamma", gamma), ("beta", beta)):
        _chk(t, torch.float32, n)
    _chk(idx, torch.int32, "idx")
    B, N, Cin = fea.shape
    S, K = idx.shape[1], idx.shape[2]
    Cout = W.shape[0]
    if W.shape[1] != Cin + 3:
        raise ValueError(f"W must be (Cout, Cin+3) = ({Cout}, {Cin + 3}), got {tuple(W.shape)}")
    out = torch.empty(B, S, Cout, device=fea.device, dtype=torch.float32)
    lib = _lib.load()
    if not generic and not fused and Cout in PROJECTED_COUT:
        W2 = W.reshape(Cout, Cin + 3)
        # a packed copy of the feature columns (rows of the Conv2d weight are Cin+3 floats: not 16-byte aligned)
        Wf = _derived("feature-columns", (W,), lambda: W2[:, :Cin].contiguous())
        if knobs.FOLD_GATHER and knobs.GEMM_BF16X3 and Cin % 32 == 0 and Cin <= knobs.BF16X3_MAX_K and radius >= knobs.FOLD_MIN_RADIUS:
            # folded form (csrc/group_mlp.hip, FOLD): the projection's epilogue adds the POINT half of the relative-coordinate
            # term, the gather subtracts the centre half and reads no coordinates.  A property of the layer (its widths and the
            # tensors' layout class), never of the row count.
            if knobs.CENTRED_GATHER:
                # LayerNorm's mean removal moved into the layer: (I - 11^T / Cout) applied to the weight and the bias (fp64, once per
                # weight version), every projected row and centre term then has zero mean over its channels by construction
                Wfc, Wrc, bc = _derived("centred-layer", (W, bias, gamma), lambda: _centred_layer(W2, bias, gamma, Cin))
                P = linear_bf16x3(fea.reshape(B * N, Cin), Wfc, bc, rank3=(xyz.reshape(B * N, 3), Wrc.data_ptr(), 3, 1.0 / float(radius)))
                if P is not None:
                    _lib.check(lib.dpm_group_gather_ln_max_centred(_ptr(P), _ptr(centers), _ptr(idx), _ptr(Wrc), 3, _ptr(gamma), _ptr(beta),
                                                                   B, N, S, K, Cout, float(radius), _ptr(out), _stream(fea)),
                               "dpm_group_gather_ln_max_centred")
                    return out
            P = linear_bf16x3(fea.reshape(B * N, Cin), Wf, bias,
                              rank3=(xyz.reshape(B * N, 3), W2.data_ptr() + 4 * Cin, Cin + 3, 1.0 / float(radius)))
            if P is not None:
                _lib.check(lib.dpm_group_gather_ln_max_folded(_ptr(P), _ptr(centers), _ptr(idx), W2.data_ptr() + 4 * Cin, Cin + 3,
                                                              _ptr(gamma), _ptr(beta), B, N, S, K, Cout, float(radius), _ptr(out),
                                                              _stream(fea)), "dpm_group_gather_ln_max_folded")
                return out
        P = linear(fea.reshape(B * N, Cin), Wf, bias)
        _lib.check(lib.dpm_group_gather_ln_max(_ptr(P), _ptr(xyz), _ptr(centers), _ptr(idx), W2.data_ptr() + 4 * Cin,
                                               Cin + 3, _ptr(gamma), _ptr(beta), B, N, S, K, Cout, float(radius),
                                               _ptr(out), _stream(fea)), "dpm_group_gather_ln_max")
        return out
    fn = lib.dpm_group_mlp_max_generic if generic else lib.dpm_group_mlp_max
    _lib.check(fn(_ptr(xyz), _ptr(fea), _ptr(centers), _ptr(idx), _ptr(W), _ptr(bias), _ptr(gamma), _ptr(beta),
                  B, N, S, K, Cin, Cout, float(radius), _ptr(out), _stream(fea)), "dpm_group_mlp_max")
    return out


def group_mlp_max_from_xyz(xyz, W0, b0, centers, idx, W, bias, gamma, beta, radius: float,
                           fused: bool = False) -> torch.Tensor:
    """First-stage SetAbstraction with the per-point input MLP (W0 (Cin,3[,1]), b0 (Cin)) folded into the gather:
    xyz (B,N,3), centers (B,S,3), idx (B,S,K), W (Cout,Cin+3[,1,1]) -> (B,S,Cout).  Raises ValueError for
    shapes the kernels do not cover (callers fall back to linear + group_mlp_max).
    Default: the projected point feature is affine in the point, P = (W_f W0) xyz + (W_f b0 + b); the two small
    weight products are made with the GEMM and the kernel evaluates P on the fly.  fused=True: the gather-GEMM kernel
    that evaluates the 16 input features per neighbour."""
    for n, t in (("xyz", xyz), ("W0", W0), ("b0", b0), ("centers", centers), ("W", W), ("bias", bias),
                 ("gamma", gamma), ("beta", beta)):
        _chk(t, torch.float32, n)
    _chk(idx, torch.int32, "idx")
    B, N, _ = xyz.shape
    S, K = idx.shape[1], idx.shape[2]
    Cin, Cout = W0.shape[0], W.shape[0]
    if W0.shape[1] != 3 or W.shape[1] != Cin + 3:
        raise ValueError("W0 must be (Cin,3) and W (Cout,Cin+3)")
    out = torch.empty(B, S, Cout, device=xyz.device, dtype=torch.float32)
    if not fused and Cout in (32, 64, 128):
        W2 = W.reshape(Cout, Cin + 3)
        A, cvec = _derived("affine-stage0", (W, W0, b0, bias), lambda: (
            linear(W2[:, :Cin], W0.reshape(Cin, 3).t().contiguous(), exact=True),                     # (Cout,3) = W_f W0
            linear(W2[:, :Cin], b0.reshape(1, Cin), residual=bias.reshape(Cout, 1), exact=True)))     # (Cout,1) = W_f b0 + b
        if knobs.CENTRED_GATHER and knobs.FOLD_GATHER:
            Ac, cc, Wrc = _derived("affine-stage0-centred", (W, W0, b0, bias, gamma), lambda: (
                (_centre_rows(A) * _gamma_sign(gamma)).contiguous(),
                (_centre_rows(cvec.reshape(Cout, 1)) * _gamma_sign(gamma)).reshape(Cout).contiguous(),
                (_centre_rows(W2[:, Cin:]) * _gamma_sign(gamma)).contiguous()))
            _lib.check(_lib.load().dpm_group_affine_ln_max_centred(_ptr(Ac), _ptr(cc), _ptr(xyz), _ptr(centers), _ptr(idx), _ptr(Wrc), 3,
                                                                   _ptr(gamma), _ptr(beta), B, N, S, K, Cout, float(radius),
                                                                   _ptr(out), _stream(xyz)), "dpm_group_affine_ln_max_centred")
            return out
        _lib.check(_lib.load().dpm_group_affine_ln_max(_ptr(A), _ptr(cvec), _ptr(xyz), _ptr(centers), _ptr(idx),
                                                       W2.data_ptr() + 4 * Cin, Cin + 3, _ptr(gamma), _ptr(beta), B, N, S,
                                                       K, Cout, float(radius), _ptr(out), _stream(xyz)),
                   "dpm_group_affine_ln_max")
        return out
    _lib.check(_lib.load().dpm_group_mlp_max_from_xyz(_ptr(xyz), _ptr(W0), _ptr(b0), _ptr(centers), _ptr(idx), _ptr(W),
                                                      _ptr(bias), _ptr(gamma), _ptr(beta), B, N, S, K, Cin, Cout,
                                                      float(radius), _ptr(out), _stream(xyz)),
               "dpm_group_mlp_max_from_xyz")
    return out




def linear(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
           residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, exact: bool = False) -> torch.Tensor:
    """x (..., Cin) (last dim contiguous rows), W (Cout, Cin[,1[,1]]) -> (..., Cout).

    `out` may be a column slice of a wider row-major buffer (its row stride is honoured).
    Which of the two GEMM kernels runs is a property of the LAYER (its Cin, Cout and the `exact` request), never of the row
    count: a frame's result does not depend on the batch it travels in.  exact=True: the fp32-MFMA kernel (callers whose
    result must equal another kernel's bit for bit: linear_layernorm's two-kernel form against its fused form)."""
    if W.dtype != torch.float32 or not W.is_cuda:
        raise ValueError("W must be an fp32 GPU tensor")
    Cout, Cin = W.shape[0], W.shape[1]
    if knobs.GEMM_BF16X3 and not exact and Cin % 32 == 0 and Cin <= knobs.BF16X3_MAX_K and Cout % 4 == 0 and W.numel() == Cout * Cin:
        done = linear_bf16x3(x, W, bias, act, residual, out)
        if done is not None:
            return done
    if W.dim() == 2 and W.stride(1) == 1 and W.stride(0) >= Cin:
        ldw = W.stride(0)      # a column range of a wider weight matrix (e.g. the feature columns of a Conv2d weight)
    else:
        _chk(W, torch.float32, "W")
        ldw = Cin
    if x.dtype != torch.float32 or not x.is_cuda or x.stride(-1) != 1:
        raise ValueError("x must be an fp32 GPU tensor with unit stride in the last dimension")
    x2 = x.reshape(-1, x.shape[-1]) if x.is_contiguous() else x
    if x2.dim() != 2:
        raise ValueError("non-contiguous x must be 2-D")
    R = x2.shape[0]
    if out is None:
        out = torch.empty(*x.shape[:-1], Cout, device=x.device, dtype=torch.float32)
    o2 = out.reshape(-1, Cout) if out.is_contiguous() else out
    if o2.dim() != 2 or o2.stride(1) != 1:
        raise ValueError("out must be 2-D with unit column stride")
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, Cout)
        _chk(r2, torch.float32, "residual")
    _lib.check(_lib.load().dpm_linear(_ptr(x2), x2.stride(0), _ptr(W), ldw, _ptr(bias), _ptr(r2),
                                      Cout if r2 is not None else 0, _ptr(o2), o2.stride(0), R, Cin, Cout, act,
                                      _stream(x)), "dpm_linear")
    return out


def _weight_planes(W: torch.Tensor):
    """The three bf16 planes (hi | mid | lo, csrc/gemm_b3.hip) of the PARAMETER a weight (view) belongs to, made once per
    weight version, and where W's first row sits in them: -> (planes (3, n) int16, element offset, plane stride n), or None
    when W is not a block of whole rows of a contiguous fp32 tensor."""
    base = W._base if W._base is not None else W
    Cin = W.shape[1]
    W2 = W if W.dim() == 2 else W.reshape(W.shape[0], Cin)
    if (base.dtype != torch.float32 or not base.is_cuda or not base.is_contiguous() or W2.stride(1) != 1 or
            W2.stride(0) != Cin or base.numel() % 8 != 0):
        return None
    off = W2.storage_offset() - base.storage_offset()
    if off < 0 or off + W2.numel() > base.numel():
        return None

    def make():
        planes = torch.empty(3, base.numel(), device=base.device, dtype=torch.int16)
        _lib.check(_lib.load().dpm_split_bf16x3(_ptr(base), base.numel(), _ptr(planes), _stream(base)), "dpm_split_bf16x3")
        return planes
    return _derived("bf16x3-planes", (base,), make), off, base.numel()


def linear_bf16x3(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
                  residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, rank3=None) -> Optional[torch.Tensor]:
    """linear() on the bf16 matrix pipe: every fp32 operand split exactly into three bf16 terms, six term products
    accumulated in fp32 (csrc/gemm_b3.hip; fp32-accumulation accuracy, not the fp32 kernel's bits).  Returns None when the
    shape / layout is not covered (the caller runs linear())."""
    Cout, Cin = W.shape[0], W.shape[1]
    if Cin % 32 != 0 or Cout % 4 != 0 or x.dtype != torch.float32 or not x.is_cuda or x.stride(-1) != 1:
        return None
    if bias is not None and bias.data_ptr() % 16:
        return None
    wp = _weight_planes(W)
    if wp is None:
        return None
    planes, off, n = wp
    x2 = x.reshape(-1, x.shape[-1]) if x.is_contiguous() else x
    if x2.dim() != 2:
        return None
    if out is None:
        out = torch.empty(*x.shape[:-1], Cout, device=x.device, dtype=torch.float32)
    o2 = out.reshape(-1, Cout) if out.is_contiguous() else out
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, Cout)
        _chk(r2, torch.float32, "residual")
    if rank3 is not None:
        # rank3 = (x3 (R,3) contiguous fp32, address of w3[0, 0], its row stride in floats, scale): out += scale * x3 w3^T in the
        # epilogue (the point half of a grouping layer's relative-coordinate term: group_mlp_max)
        x3, w3_ptr, ldw3, scale = rank3
        _chk(x3, torch.float32, "x3")
        if tuple(x3.shape) != (x2.shape[0], 3):
            raise ValueError("rank3 rows must be (R, 3)")
        st = _lib.load().dpm_linear_bf16x3_rank3(_ptr(x2), x2.stride(0), planes.data_ptr() + 2 * off, Cin, n, _ptr(bias), _ptr(r2),
                                                 Cout if r2 is not None else 0, _ptr(o2), o2.stride(0), x2.shape[0], Cin, Cout, act,
                                                 _ptr(x3), ctypes.c_void_p(w3_ptr), int(ldw3), float(scale), _stream(x))
    else:
        st = _lib.load().dpm_linear_bf16x3(_ptr(x2), x2.stride(0), planes.data_ptr() + 2 * off, Cin, n, _ptr(bias), _ptr(r2),
                                           Cout if r2 is not None else 0, _ptr(o2), o2.stride(0), x2.shape[0], Cin, Cout, act,
                                           _stream(x))
    if st == -2:
        return None
    _lib.check(st, "dpm_linear_bf16x3")
    return out


def similarity_batched(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a (B,M,C), b (B,N,C) contiguous -> S (B,M,N) with S[p] = a[p] @ b[p]^T (fp32 MFMA)."""
    _chk(a, torch.float32, "a"), _chk(b, torch.float32, "b")
    B, M, C = a.shape
    N = b.shape[1]
    out = torch.empty(B, M, N, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().dpm_linear_batched(_ptr(a), C, M * C, _ptr(b), C, N * C, None, None, 0, 0, _ptr(out), N,
                                              M * N, B, M, C, N, ACT_NONE, _stream(a)), "dpm_linear_batched")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, act: int = ACT_NONE,
              pre: Optional[torch.Tensor] = None, post: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(LN(x + pre) * gamma + beta + post) over the last dimension."""
    _chk(x, torch.float32, "x")
    C = x.shape[-1]
    x2 = x.reshape(-1, C)
    out = torch.empty_like(x)
    for n, t in (("pre", pre), ("post", post)):
        if t is not None:
            _chk(t, torch.float32, n)
            if t.numel() != x.numel():
                raise ValueError(f"{n} must have the shape of x")
    _lib.check(_lib.load().dpm_layernorm(_ptr(x2), C, _ptr(pre), _ptr(gamma), _ptr(beta), _ptr(post), _ptr(out), C,
                                         x2.shape[0], C, act, _stream(x)), "dpm_layernorm")
    return out


FUSED_LN_WIDTHS = (32, 64, 128, 256)  # dpm_linear_layernorm: the output row must fit one workgroup's tile
FUSED_LN_MIN_ROWS = 16384             # ... and there must be a workgroup (64 rows) for every compute unit


def linear_layernorm(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                     act: int = ACT_NONE, pre: Optional[torch.Tensor] = None, post: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(LN(x W^T + bias + pre) * gamma + beta + post): Conv1d(k=1)/Linear + LayerNorm1d in one kernel when the
    output width is one of FUSED_LN_WIDTHS, otherwise the GEMM kernel followed by the LayerNorm kernel (same result up
    to the summation order of the row statistics)."""
    Cout, Cin = W.shape[0], W.shape[1]
    x2 = x.reshape(-1, x.shape[-1])
    # Layers the bf16x3 kernel covers (a property of the layer: K <= 512 in whole K-tiles, whole weight rows) take it in BOTH forms
    # -- fused from FUSED_LN_MIN_ROWS rows on, GEMM + LayerNorm below -- with identical rows either way
    # (from K = 128 on: with shorter reductions the kernels are bound by their epilogues and the fp32 form's smaller row tiles win)
    if (knobs.GEMM_BF16X3 and knobs.GEMM_LN_BF16X3 and Cin % 32 == 0 and knobs.BF16X3_LN_MIN_K <= Cin <= knobs.BF16X3_MAX_K and Cout % 4 == 0 and W.numel() == Cout * Cin
            and x.dtype == torch.float32):
        wp = _weight_planes(W)
        if wp is not None and (bias is None or bias.data_ptr() % 16 == 0):
            if (Cout in FUSED_LN_WIDTHS and (x2.shape[0] >= FUSED_LN_MIN_ROWS or x2.shape[0] <= knobs.FUSED_LN_SMALL_ROWS)
                    and x2.is_contiguous() and knobs.FUSED_LN):
                planes, off, n = wp
                out = torch.empty(*x.shape[:-1], Cout, device=x.device, dtype=torch.float32)
                for nm, t in (("pre", pre), ("post", post)):
                    if t is not None:
                        _chk(t, torch.float32, nm)
                        if t.numel() != out.numel():
                            raise ValueError(f"{nm} must have the shape of the output")
                st = _lib.load().dpm_linear_layernorm_bf16x3(_ptr(x2), x2.stride(0), planes.data_ptr() + 2 * off, Cin, n, _ptr(bias),
                                                             _ptr(pre), _ptr(gamma), _ptr(beta), _ptr(post), _ptr(out), Cout,
                                                             x2.shape[0], Cin, Cout, act, _stream(x))
                if st == 0:
                    return out
                if st != -2:
                    _lib.check(st, "dpm_linear_layernorm_bf16x3")
            y = linear_bf16x3(x, W, bias, residual=pre)
            if y is not None:
                return layernorm(y, gamma, beta, act=act, post=post)
    # the fused kernel owns whole output rows (64 rows per workgroup): below FUSED_LN_MIN_ROWS it leaves most of the 256
    # compute units idle and the two-kernel form is 1.5-2.8x faster (scripts/gemm_ln_shapes.py: 4096 x 1024 -> 256 takes
    # 86 us fused, 31 us as GEMM + LayerNorm)
    if (Cout in FUSED_LN_WIDTHS and x2.shape[0] >= FUSED_LN_MIN_ROWS and W.is_contiguous() and x2.is_contiguous()
            and x.dtype == torch.float32 and Cin % 4 == 0 and knobs.FUSED_LN):
        _chk(W, torch.float32, "W")
        out = torch.empty(*x.shape[:-1], Cout, device=x.device, dtype=torch.float32)
        for n, t in (("pre", pre), ("post", post)):
            if t is not None:
                _chk(t, torch.float32, n)
                if t.numel() != out.numel():
                    raise ValueError(f"{n} must have the shape of the output")
        st = _lib.load().dpm_linear_layernorm(_ptr(x2), x2.stride(0), _ptr(W), Cin, _ptr(bias), _ptr(pre), _ptr(gamma),
                                              _ptr(beta), _ptr(post), _ptr(out), Cout, x2.shape[0], Cin, Cout, act, _stream(x))
        if st == 0:
            return out
        if st != -2:  # anything but "unsupported shape"
            _lib.check(st, "dpm_linear_layernorm")
    y = linear(x, W, bias, residual=pre, exact=True)   # the fused kernel's arithmetic: same bits in either form
    return layernorm(y, gamma, beta, act=act, post=post)


def _pwconv_kperm() -> torch.Tensor:
    """stored column -> original column of W2 for dpm_pwconv_pair_bf16x3 (include/dpm_hip.h): 32 s + 8 g + e <- 16 (2 s + (e >> 2)) + 4 g + (e & 3)"""
    return torch.tensor([16 * (2 * s + (e >> 2)) + 4 * g + (e & 3) for s in range(4) for g in range(4) for e in range(8)], dtype=torch.long)


def pwconv_pair(x: torch.Tensor, W1, b1, g1, be1, W2, b2, g2, be2, post: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """relu(LN2(relu(LN1(x W1^T + b1)) W2^T + b2) + post) in one kernel (InvResMLP's pw_conv pair, C = 32 only: csrc/gemm_b3.hip,
    pwconv_pair_b3_kernel).  None when the layer / layout is not covered: the caller runs two linear_layernorm calls -- for EVERY
    call of that layer (the decision depends on the layer's shape and the tensors' layout class, never on the row count)."""
    C, H = W1.shape[1], W1.shape[0]
    if not (knobs.FUSED_PWCONV and knobs.GEMM_BF16X3 and C == 32 and H == 128 and tuple(W2.shape[:2]) == (C, H) and x.is_cuda
            and x.dtype == torch.float32 and x.is_contiguous() and W1.numel() == C * H and W2.numel() == C * H):
        return None
    wp1 = _weight_planes(W1)
    if wp1 is None or (post is not None and (not post.is_contiguous() or post.numel() != x.numel())):
        return None
    for n, t_ in (("gamma1", g1), ("beta1", be1), ("gamma2", g2), ("beta2", be2)):
        _chk(t_, torch.float32, n)

    def make():
        Wp = W2.reshape(C, H)[:, _pwconv_kperm().to(W2.device)].contiguous()
        planes = torch.empty(3, C * H, device=W2.device, dtype=torch.int16)
        _lib.check(_lib.load().dpm_split_bf16x3(_ptr(Wp), C * H, _ptr(planes), _stream(W2)), "dpm_split_bf16x3")
        planes._dpm_keep = Wp   # (the split kernel reads it asynchronously)
        return planes
    planes2 = _derived("bf16x3-planes-kperm", (W2,), make)
    planes1, off1, n1 = wp1
    x2 = x.reshape(-1, C)
    out = torch.empty_like(x)
    st = _lib.load().dpm_pwconv_pair_bf16x3(_ptr(x2), C, planes1.data_ptr() + 2 * off1, n1, _ptr(b1), _ptr(g1), _ptr(be1),
                                            planes2.data_ptr(), C * H, _ptr(b2), _ptr(g2), _ptr(be2), _ptr(post), _ptr(out),
                                            x2.shape[0], C, H, _stream(x))
    if st == -2:
        return None
    _lib.check(st, "dpm_pwconv_pair_bf16x3")
    return out


def three_interp_cat(xyz1, xyz2, lengths2, fea1, fea2) -> torch.Tensor:
    """fine xyz1 (B,N,3)/fea1 (B,N,D1), coarse xyz2 (B,S,3)/fea2 (B,S,D2) -> (B,N,D1+D2)."""
    for n, t in (("xyz1", xyz1), ("xyz2", xyz2), ("fea1", fea1), ("fea2", fea2)):
        _chk(t, torch.float32, n)
    _chk(lengths2, torch.int32, "lengths2")
    B, N, D1 = fea1.shape
    S, D2 = fea2.shape[1], fea2.shape[2]
    out = torch.empty(B, N, D1 + D2, device=fea1.device, dtype=torch.float32)
    _lib.check(_lib.load().dpm_three_interp_cat(_ptr(xyz1), _ptr(xyz2), _ptr(lengths2), _ptr(fea1), _ptr(fea2),
                                                B, N, S, D1, D2, _ptr(out), _stream(fea1)), "dpm_three_interp_cat")
    return out


# ---------------------------------------------------------------------------------------- decoder
def _rows2d(t: torch.Tensor, name: str) -> torch.Tensor:
    """2-D fp32 GPU view with unit column stride (row stride free)."""
    if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D fp32 GPU tensor with unit column stride")
    return t


def posemb(xyz_rows: torch.Tensor, dim_t: torch.Tensor, emb_dim: int) -> torch.Tensor:
    """xyz_rows (R,3) view (row stride free) -> (R, emb_dim)."""
    _rows2d(xyz_rows, "xyz")
    _chk(dim_t, torch.float32, "dim_t")
    R = xyz_rows.shape[0]
    out = torch.empty(R, emb_dim, device=xyz_rows.device, dtype=torch.float32)
    _lib.check(_lib.load().dpm_posemb(_ptr(xyz_rows), xyz_rows.stride(0), _ptr(dim_t), dim_t.numel(), emb_dim, R,
                                      _ptr(out), _stream(out)), "dpm_posemb")
    return out


ATTENTION_SPLIT_MIN_KEYS = 1024   # key-split attention: from this many keys on ...
ATTENTION_SPLIT_BLOCKS = 256      # ... while ONE sequence's plain launch would leave the chip's CUs without a workgroup each


def attention_key_splits(B: int, M: int, N: int, heads: int, head_dim: int) -> int:
    """Number of key ranges for dpm_attention_split (1 = plain kernel): few queries against many keys -- a scan's tokens
    attending a map tile -- give a handful of workgroups that each walk the whole key sequence.  Depends on the shape of
    ONE sequence only, never on the batch size, so a pair's result does not depend on the batch it travels in."""
    if head_dim != 32 or N < ATTENTION_SPLIT_MIN_KEYS:
        return 1
    blocks = -(-M // 64) * heads   # per sequence: B stays out of the rule (see above)
    if blocks >= ATTENTION_SPLIT_BLOCKS:
        return 1
    ns = max(1, min(ATTENTION_SPLIT_BLOCKS // blocks, N // 256, 64))
    chunk = -(-N // (64 * ns)) * 64   # keys per range (whole tiles); drop ranges that would start beyond the last key
    return -(-N // chunk)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, M: int, N: int, heads: int = 8,
              out: Optional[torch.Tensor] = None, kv_shift: int = 0, key_mask: Optional[torch.Tensor] = None,
              seq_index: Optional[torch.Tensor] = None):
    """q (B*M,E) / k,v (B*N,E) row views (column slices of wider buffers allowed) -> (B*M,E)
    (written into `out`, a contiguous (B*M,E) tensor or row range of one, when given).
    kv_shift: sequence b attends the keys / values of sequence (b + kv_shift) mod B.
    key_mask (B,N) uint8, non-zero = padding key (nn.MultiheadAttention's key_padding_mask), indexed like the keys.
    seq_index (B,) int32: q / k / v hold U stored sequences ((U*M,E) / (U*N,E)) and batch element b is stored sequence
    seq_index[b] (its keys / values: stored sequence seq_index[(b + kv_shift) mod B]); the output stays (B*M,E)."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _rows2d(t, n)
    E = q.shape[1]
    if out is None:
        out = torch.empty(B * M, E, device=q.device, dtype=torch.float32)
    elif tuple(out.shape) != (B * M, E) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous fp32 (B*M, E) tensor")
    if key_mask is not None:
        _chk(key_mask, torch.uint8, "key_mask")
        if tuple(key_mask.shape) != (B, N):
            raise ValueError(f"key_mask must be ({B}, {N}), got {tuple(key_mask.shape)}")
    if seq_index is not None:
        _chk(seq_index, torch.int32, "seq_index")
        if seq_index.numel() != B or key_mask is not None:
            raise ValueError("seq_index must hold B entries and excludes key_mask")
        _lib.check(_lib.load().dpm_attention_indexed(_ptr(q), q.stride(0), M * q.stride(0), _ptr(k), k.stride(0),
                                                     N * k.stride(0), _ptr(v), v.stride(0), N * v.stride(0), _ptr(out), E,
                                                     M * E, B, M, N, heads, E // heads, int(kv_shift), _ptr(seq_index),
                                                     _stream(q)), "dpm_attention_indexed")
        return out
    nsplit = attention_key_splits(B, M, N, heads, E // heads) if key_mask is None else 1
    if nsplit > 1:
        lib = _lib.load()
        ws = torch.empty(lib.dpm_attention_split_workspace_bytes(B, M, heads, E // heads, nsplit), device=q.device, dtype=torch.uint8)
        _lib.check(lib.dpm_attention_split(_ptr(q), q.stride(0), M * q.stride(0), _ptr(k), k.stride(0), N * k.stride(0),
                                           _ptr(v), v.stride(0), N * v.stride(0), _ptr(out), E, M * E, B, M, N, heads,
                                           E // heads, int(kv_shift), nsplit, _ptr(ws), _stream(q)), "dpm_attention_split")
        return out
    _lib.check(_lib.load().dpm_attention_masked(_ptr(q), q.stride(0), M * q.stride(0), _ptr(k), k.stride(0),
                                                N * k.stride(0), _ptr(v), v.stride(0), N * v.stride(0), _ptr(out), E,
                                                M * E, B, M, N, heads, E // heads, int(kv_shift), _ptr(key_mask), _stream(q)),
               "dpm_attention")
    return out


# fewer rows than this: linear() + attention() (identical results; one pair's 512 rows are pure latency, and there the planes'
# epilogue costs more than the split it saves: replayed graph 0.46 -> 0.48 ms)
KV_PLANES_MIN_ROWS = 2048


def linear_kvplanes(x: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, M: int, heads: int = 8):
    """The q | k | v projection half of qkv_attention: x (U*M, E) -> (q (U*M, E) fp32 rows, kv planes (uint8, one 24 KB image per
    (sequence, head, 64-key tile))), or None when the shape is not covered (dpm_linear_bf16x3_kvplanes)."""
    E = x.shape[-1]
    if (not knobs.KV_PLANES or not knobs.GEMM_BF16X3 or E != heads * 32 or M % 64 != 0 or tuple(W.shape) != (3 * E, E) or
            E > knobs.BF16X3_MAX_K or x.dim() != 2 or x.shape[0] % M != 0 or x.dtype != torch.float32 or not x.is_cuda or
            x.stride(1) != 1 or bias is None or bias.data_ptr() % 16 or x.shape[0] < KV_PLANES_MIN_ROWS):
        return None
    wp = _weight_planes(W)
    if wp is None:
        return None
    planes, off, n = wp
    U = x.shape[0] // M
    lib = _lib.load()
    q = torch.empty(U * M, E, device=x.device, dtype=torch.float32)
    kv = torch.empty(lib.dpm_attention_planes_bytes(U, M, heads), device=x.device, dtype=torch.uint8)
    st = lib.dpm_linear_bf16x3_kvplanes(_ptr(x), x.stride(0), planes.data_ptr() + 2 * off, E, n, _ptr(bias), _ptr(q), E, U * M, E, 3 * E, E,
                                        M, heads, _ptr(kv), _stream(x))
    if st == -2:
        return None
    _lib.check(st, "dpm_linear_bf16x3_kvplanes")
    return q, kv


def attention_planes(q: torch.Tensor, kv: torch.Tensor, B: int, M: int, heads: int = 8, kv_shift: int = 0,
                     seq_index: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The attention half of qkv_attention: q (U*M, E) rows and the planes linear_kvplanes made of K / V -> (B*M, E)."""
    E = q.shape[1]
    out = torch.empty(B * M, E, device=q.device, dtype=torch.float32)
    _lib.check(_lib.load().dpm_attention_planes(_ptr(q), E, M * E, _ptr(kv), _ptr(out), E, M * E, B, M, M, heads, int(kv_shift),
                                                _ptr(seq_index), _stream(q)), "dpm_attention_planes")
    return out


def qkv_attention(x: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, B: int, M: int, heads: int = 8, kv_shift: int = 0,
                  seq_index: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """attention(q, k, v) with q | k | v = x W^T + bias in one hand-over: x (U*M, E) holds U stored sequences of M tokens, W (3E, E)
    is nn.MultiheadAttention's in_proj_weight; batch element b is stored sequence seq_index[b] (None: b, U == B) and attends the
    keys / values of element (b + kv_shift) mod B.  The projection writes Q as fp32 rows and K / V as the attention kernel's bf16
    operand planes (dpm_linear_bf16x3_kvplanes -> dpm_attention_planes): the same values split the same way as
    linear() + attention() would, once per key tile instead of once per query block -- identical results.
    Returns None when the shape is not covered (32-wide heads, M % 64 == 0, no mask, no key ranges, at least KV_PLANES_MIN_ROWS rows;
    the caller runs linear() + attention())."""
    if x.dim() != 2 or M <= 0 or x.shape[0] % M != 0 or attention_key_splits(B, M, M, heads, 32) > 1:
        return None
    if seq_index is not None:
        _chk(seq_index, torch.int32, "seq_index")
        if seq_index.numel() != B:
            raise ValueError("seq_index must hold B entries")
    elif x.shape[0] // M != B:
        raise ValueError("x must hold B sequences when seq_index is not given")
    made = linear_kvplanes(x, W, bias, M, heads)
    if made is None:
        return None
    return attention_planes(made[0], made[1], B, M, heads, kv_shift, seq_index)


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    _chk(x, torch.float32, "x")
    out = torch.empty_like(x)
    _lib.check(_lib.load().dpm_l2_normalize(_ptr(x), x.numel() // x.shape[-1], x.shape[-1], _ptr(out), _stream(x)),
               "dpm_l2_normalize")
    return out


def dual_softmax_topk(S: torch.Tensor, tau: float, k: int):
    """S (M,N) or (B,M,N) similarity (overwritten with the dual-softmax matrix) ->
    (values (k,) / (B,k), flat idx int32 of the same shape), each row sorted descending."""
    _chk(S, torch.float32, "S")
    single = S.dim() == 2
    B = 1 if single else S.shape[0]
    M, N = S.shape[-2], S.shape[-1]
    lib = _lib.load()
    val = torch.empty(B, k, device=S.device, dtype=torch.float32)
    idx = torch.empty(B, k, device=S.device, dtype=torch.int32)
    ws = torch.empty(lib.dpm_pairing_workspace_bytes(B, M, N), device=S.device, dtype=torch.uint8)
    _lib.check(lib.dpm_dual_softmax_topk(_ptr(S), B, M, N, float(tau), k, _ptr(val), _ptr(idx), _ptr(ws), _stream(S)),
               "dpm_dual_softmax_topk")
    return (val[0], idx[0]) if single else (val, idx)


MATCH_MAX_N, MATCH_MAX_K = 256, 2048   # dpm_match_topk: columns one workgroup holds, pairs its lists hold
MATCH_MAX_MERGE = 8192                 # candidates (strips x k) the last workgroup of a pair merges


def match_supported(M: int, N: int, C: int, k: int) -> bool:
    """shapes dpm_match_topk takes (a function of ONE pair's shape, never of the batch)"""
    # ... and few enough row strips that ONE workgroup merges their candidates quickly: a 4096 x 256 map tile (64 strips x 1088
    # candidates) spent 485 us in that merge against ~200 us for the whole five-kernel form
    strips = -(-M // 64)
    return (N <= MATCH_MAX_N and k <= MATCH_MAX_K and C % 32 == 0 and 1 <= k <= M * N and strips * min(k, 64 * N) <= MATCH_MAX_MERGE)


def match_topk(a: torch.Tensor, b: torch.Tensor, tau: float, k: int):
    """a (B,M,C), b (B,N,C) L2-normalised head outputs -> (values (B,k), flat idx (B,k) int32), sorted descending: similarity,
    dual softmax and top-k of decoder.py:185-191 without the (M,N) matrix in memory.  Raises ValueError for shapes outside
    match_supported()."""
    _chk(a, torch.float32, "a"), _chk(b, torch.float32, "b")
    B, M, C = a.shape
    N = b.shape[1]
    lib = _lib.load()
    val = torch.empty(B, k, device=a.device, dtype=torch.float32)
    idx = torch.empty(B, k, device=a.device, dtype=torch.int32)
    ws = torch.empty(lib.dpm_match_workspace_bytes(B, M, N, k), device=a.device, dtype=torch.uint8)
    _lib.check(lib.dpm_match_topk(_ptr(a), _ptr(b), B, M, N, C, float(tau), k, _ptr(val), _ptr(idx), _ptr(ws), _stream(a)),
               "dpm_match_topk")
    return val, idx


def gather_pairs(x: torch.Tensor, y: torch.Tensor, flat: torch.Tensor):
    """x (B,M,E), y (B,N,E), flat (B,k) -> X (B,2k,2E), src_idx (B,k), dst_idx (B,k)   (2-D inputs: B = 1, 2-D outputs)."""
    _chk(x, torch.float32, "x"), _chk(y, torch.float32, "y"), _chk(flat, torch.int32, "flat")
    single = x.dim() == 2
    B = 1 if single else x.shape[0]
    M, E, N, k = x.shape[-2], x.shape[-1], y.shape[-2], flat.shape[-1]
    X = torch.empty(B, 2 * k, 2 * E, device=x.device, dtype=torch.float32)
    si = torch.empty(B, k, device=x.device, dtype=torch.int32)
    di = torch.empty(B, k, device=x.device, dtype=torch.int32)
    _lib.check(_lib.load().dpm_gather_pairs(_ptr(x), _ptr(y), _ptr(flat), B, k, M, N, E, _ptr(X), _ptr(si), _ptr(di),
                                            _stream(x)), "dpm_gather_pairs")
    return (X[0], si[0], di[0]) if single else (X, si, di)


def mean_rows(x: torch.Tensor, out: torch.Tensor) -> None:
    """x (B,R,C) -> out (B,C) view (row stride free)."""
    _chk(x, torch.float32, "x")
    _rows2d(out, "out")
    B, R, C = x.shape
    _lib.check(_lib.load().dpm_mean_rows(_ptr(x), B, R, C, _ptr(out), out.stride(0), _stream(x)), "dpm_mean_rows")


RES_HDR = 20  # floats before the inlier-confidence list in a corr_kabsch result


def corr_kabsch(offsets, src_xyz, dst_xyz, src_idx, dst_idx, conf, eps_offset: float, num_iter: int = 3,
                std_ratio: float = 3.0, header_out: Optional[torch.Tensor] = None, batch: int = 1) -> torch.Tensor:
    """-> result (batch, 20 + 2k) fp32 (1-D when batch == 1 and conf is 1-D): R(9) T(3) rmse n_corr n_inlier iters
    conf30 (3 reserved), then the inlier confidences.  src_xyz / dst_xyz: (batch*M, 3) row views (row stride
    free) holding the batch elements back to back.  offsets None: (conf, src_xyz, dst_xyz) are ready-made
    correspondences.  header_out: optional (batch, >=20) fp32 view (unit column stride) that also receives
    result[:, :20]."""
    _chk(conf, torch.float32, "conf")
    _rows2d(src_xyz, "src_xyz"), _rows2d(dst_xyz, "dst_xyz")
    k = conf.shape[-1]
    if offsets is not None:
        _chk(offsets, torch.float32, "offsets")
        _chk(src_idx, torch.int32, "src_idx"), _chk(dst_idx, torch.int32, "dst_idx")
    lib = _lib.load()
    ws = torch.empty(lib.dpm_kabsch_workspace_bytes(batch, k), device=conf.device, dtype=torch.uint8)
    result = torch.empty(batch, RES_HDR + 2 * k, device=conf.device, dtype=torch.float32)
    hs = 0
    if header_out is not None:
        if header_out.dim() == 1:
            header_out = header_out.unsqueeze(0)
        _rows2d(header_out, "header_out")
        hs = header_out.stride(0)
    Ms, Md = src_xyz.shape[0] // batch, dst_xyz.shape[0] // batch
    _lib.check(lib.dpm_corr_kabsch(_ptr(offsets), _ptr(src_xyz), src_xyz.stride(0), Ms * src_xyz.stride(0),
                                   _ptr(dst_xyz), dst_xyz.stride(0), Md * dst_xyz.stride(0), _ptr(src_idx),
                                   _ptr(dst_idx), _ptr(conf), batch, k, float(eps_offset), num_iter, float(std_ratio),
                                   _ptr(ws), _ptr(result), _ptr(header_out), hs, _stream(conf)), "dpm_corr_kabsch")
    return result[0] if (batch == 1 and conf.dim() == 1) else result


def information_matrix(pcd1: torch.Tensor, pcd2: torch.Tensor, Rt: torch.Tensor, radius: float = 1.0,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """pcd1 (3,N1), pcd2 (3,N2) metres on the GPU, Rt (>=12,) [R row-major, T, ...] -> (6,6) fp32 on the GPU
    (written into `out`, a contiguous 36-float view, when given)."""
    _chk(pcd1, torch.float32, "pcd1"), _chk(pcd2, torch.float32, "pcd2"), _chk(Rt, torch.float32, "Rt")
    N1, N2 = pcd1.shape[1], pcd2.shape[1]
    lib = _lib.load()
    ws = torch.empty(lib.dpm_infomat_workspace_bytes(1, N1, N2), device=pcd1.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty(6, 6, device=pcd1.device, dtype=torch.float32)
    else:
        _chk(out, torch.float32, "out")
    _lib.check(lib.dpm_information_matrix(_ptr(pcd1), N1, _ptr(pcd2), N2, _ptr(Rt), float(radius), _ptr(out),
                                          _ptr(ws), _stream(pcd1)), "dpm_information_matrix")
    return out


def information_matrix_batched(pcd: torch.Tensor, src_frame: torch.Tensor, dst_frame: torch.Tensor, Rt_rows: torch.Tensor,
                               out_rows: torch.Tensor, radius: float = 1.0, grids: Optional[torch.Tensor] = None) -> None:
    """pcd (F,3,N) metres; pair p = (src_frame[p], dst_frame[p]) (int32 GPU tensors); Rt_rows (P, >=12) and
    out_rows (P, >=36) are row views (unit column stride) -- typically columns of one edge table.
    grids: the workspace returned by information_matrix_grids(pcd, dst_frame, radius) -- only the search runs."""
    _chk(pcd, torch.float32, "pcd"), _chk(src_frame, torch.int32, "src_frame"), _chk(dst_frame, torch.int32, "dst_frame")
    _rows2d(Rt_rows, "Rt_rows"), _rows2d(out_rows, "out_rows")
    P_, N = src_frame.numel(), pcd.shape[2]
    lib = _lib.load()
    if grids is not None:
        if grids.numel() != lib.dpm_infomat_workspace_bytes(P_, N, N):
            raise ValueError("grids was built for a different (n_pairs, N)")
        _lib.check(lib.dpm_infomat_search_grids(_ptr(pcd), N, _ptr(src_frame), _ptr(dst_frame), P_, _ptr(Rt_rows),
                                                Rt_rows.stride(0), float(radius), _ptr(out_rows), out_rows.stride(0),
                                                _ptr(grids), _stream(pcd)), "dpm_infomat_search_grids")
        return
    ws = torch.empty(lib.dpm_infomat_workspace_bytes(P_, N, N), device=pcd.device, dtype=torch.uint8)
    _lib.check(lib.dpm_information_matrix_batched(_ptr(pcd), N, _ptr(src_frame), _ptr(dst_frame), P_, _ptr(Rt_rows),
                                                  Rt_rows.stride(0), float(radius), _ptr(out_rows), out_rows.stride(0),
                                                  _ptr(ws), _stream(pcd)), "dpm_information_matrix_batched")


def information_matrix_grids(pcd: torch.Tensor, dst_frame: torch.Tensor, radius: float = 1.0) -> torch.Tensor:
    """The pose-independent half of information_matrix_batched: sorts the target scan of every pair into its
    search grid.  Returns the workspace to pass as `grids=` (same pcd, dst_frame and radius)."""
    _chk(pcd, torch.float32, "pcd"), _chk(dst_frame, torch.int32, "dst_frame")
    P_, N = dst_frame.numel(), pcd.shape[2]
    lib = _lib.load()
    ws = torch.empty(lib.dpm_infomat_workspace_bytes(P_, N, N), device=pcd.device, dtype=torch.uint8)
    _lib.check(lib.dpm_infomat_build_grids(_ptr(pcd), N, _ptr(dst_frame), P_, float(radius), _ptr(ws), _stream(pcd)),
               "dpm_infomat_build_grids")
    return ws
