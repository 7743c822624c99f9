"""ctypes binding of libdpm_hip.so (the C ABI declared in include/dpm_hip.h).

There is NO fallback: if the library is missing or a call fails, this module raises.  ctypes
releases the GIL for the duration of every foreign call, so the SLAM threads of the reference
(system/core.py:82-109) can drive the kernels concurrently.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  -- must precede the CDLL below: libdpm_hip.so has to bind to the HIP runtime torch ships
from ctypes import c_char_p, c_double, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DPM_LIB (read once, at import): another build of the library for A/B measurements (csrc/build.py --out).  bench.py refuses
# to run with it unless --allow-knobs is given; no entry point of the shipped library reads the environment.
LIB_PATH = os.environ.get("DPM_LIB") or os.path.join(_HERE, "libdpm_hip.so")
VERSION_EXPERIMENT = 0x40000000   # include/dpm_hip.h: DPM_VERSION_EXPERIMENT

P, I, D, LL = c_void_p, c_int, c_double, c_longlong

# name -> (restype, argtypes); mirrors include/dpm_hip.h one to one (tests check the symbol list)
SIGNATURES = {
    "dpm_version": (I, []),
    "dpm_error_string": (c_char_p, [I]),
    "dpm_prepare_points": (I, [P, P, I, I, I, P, P, P]),
    "dpm_to_channel_first": (I, [P, I, I, I, P, P]),
    "dpm_to_channel_first_ld": (I, [P, I, I, I, P, I, P]),
    "dpm_emit_descriptors": (I, [P, P, P, I, I, I, D, P, P, P, P, P]),
    "dpm_nested_levels": (I, [P, P, I, I, I, P, P, P, P, P]),
    "dpm_gather_frames": (I, [P, LL, I, I, I, P, I, P, P]),
    "dpm_fps_workspace_bytes": (c_size_t, [I, I, I]),
    "dpm_fps": (I, [P, P, I, I, I, P, P, P, P, P]),
    "dpm_fps_ex": (I, [P, P, I, I, I, P, P, P, P, I, P]),
    "dpm_fps_start": (I, [P, P, P, I, I, I, P, P, P, P, P]),
    "dpm_knn_workspace_bytes": (c_size_t, [I, I]),
    "dpm_knn_hybrid": (I, [P, P, P, I, I, I, I, D, P, P, P]),
    "dpm_knn_hybrid_reuse": (I, [P, P, P, I, I, I, I, D, P, P, P, P, P]),
    "dpm_knn_build_grid": (I, [P, P, I, I, D, P, P]),
    "dpm_knn_hybrid_prebuilt": (I, [P, P, P, I, I, I, I, D, P, P, P]),
    "dpm_ball_query": (I, [P, P, P, I, I, I, I, D, P, P]),
    "dpm_voxel_sampler_bounds": (I, [P, P, I, I, I, D, D, P, P]),
    "dpm_voxel_sampler_workspace_bytes": (c_size_t, [I, I, LL]),
    "dpm_voxel_sampler_select": (I, [P, P, I, I, I, D, D, P, LL, I, P, I, P, P, P]),
    "dpm_posegraph_optimize": (I, [P, I, P, P, P, P, I, I, P, P, P]),
    "dpm_host_topk_replay": (I, [P, I, I, I, P]),
    "dpm_host_sort_replay": (I, [P, I, I, P]),
    "dpm_group_mlp_max": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, D, P, P]),
    "dpm_group_mlp_max_generic": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, D, P, P]),
    "dpm_group_mlp_max_from_xyz": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, D, P, P]),
    "dpm_group_gather_ln_max": (I, [P, P, P, P, P, I, P, P, I, I, I, I, I, D, P, P]),
    "dpm_group_gather_ln_max_folded": (I, [P, P, P, P, I, P, P, I, I, I, I, I, D, P, P]),
    "dpm_group_affine_ln_max": (I, [P, P, P, P, P, P, I, P, P, I, I, I, I, I, D, P, P]),
    "dpm_group_gather_ln_max_centred": (I, [P, P, P, P, I, P, P, I, I, I, I, I, D, P, P]),
    "dpm_group_affine_ln_max_centred": (I, [P, P, P, P, P, P, I, P, P, I, I, I, I, I, D, P, P]),
    "dpm_linear": (I, [P, I, P, I, P, P, I, P, I, I, I, I, I, P]),
    "dpm_linear_batched": (I, [P, I, LL, P, I, LL, P, P, I, LL, P, I, LL, I, I, I, I, I, P]),
    "dpm_split_bf16x3": (I, [P, LL, P, P]),
    "dpm_linear_bf16x3": (I, [P, I, P, I, LL, P, P, I, P, I, I, I, I, I, P]),
    "dpm_linear_bf16x3_rank3": (I, [P, I, P, I, LL, P, P, I, P, I, I, I, I, I, P, P, I, D, P]),
    "dpm_linear_layernorm_bf16x3": (I, [P, I, P, I, LL, P, P, P, P, P, P, I, I, I, I, I, P]),
    "dpm_pwconv_pair_bf16x3": (I, [P, I, P, LL, P, P, P, P, LL, P, P, P, P, P, I, I, I, P]),
    "dpm_layernorm": (I, [P, I, P, P, P, P, P, I, I, I, I, P]),
    "dpm_linear_layernorm": (I, [P, I, P, I, P, P, P, P, P, P, I, I, I, I, I, P]),
    "dpm_three_interp_cat": (I, [P, P, P, P, P, I, I, I, I, I, P, P]),
    "dpm_posemb": (I, [P, I, P, I, I, I, P, P]),
    "dpm_attention": (I, [P, I, LL, P, I, LL, P, I, LL, P, I, LL, I, I, I, I, I, P]),
    "dpm_attention_shifted": (I, [P, I, LL, P, I, LL, P, I, LL, P, I, LL, I, I, I, I, I, I, P]),
    "dpm_attention_masked": (I, [P, I, LL, P, I, LL, P, I, LL, P, I, LL, I, I, I, I, I, I, P, P]),
    "dpm_attention_indexed": (I, [P, I, LL, P, I, LL, P, I, LL, P, I, LL, I, I, I, I, I, I, P, P]),
    "dpm_attention_split_workspace_bytes": (c_size_t, [I, I, I, I, I]),
    "dpm_attention_planes_bytes": (c_size_t, [I, I, I]),
    "dpm_attention_planes": (I, [P, I, LL, P, P, I, LL, I, I, I, I, I, P, P]),
    "dpm_linear_bf16x3_kvplanes": (I, [P, I, P, I, LL, P, P, I, I, I, I, I, I, I, P, P]),
    "dpm_attention_split": (I, [P, I, LL, P, I, LL, P, I, LL, P, I, LL, I, I, I, I, I, I, I, P, P]),
    "dpm_l2_normalize": (I, [P, I, I, P, P]),
    "dpm_match_workspace_bytes": (c_size_t, [I, I, I, I]),
    "dpm_match_topk": (I, [P, P, I, I, I, I, D, I, P, P, P, P]),
    "dpm_pairing_workspace_bytes": (c_size_t, [I, I, I]),
    "dpm_dual_softmax_topk": (I, [P, I, I, I, D, I, P, P, P, P]),
    "dpm_gather_pairs": (I, [P, P, P, I, I, I, I, I, P, P, P, P]),
    "dpm_mean_rows": (I, [P, I, I, I, P, I, P]),
    "dpm_kabsch_workspace_bytes": (c_size_t, [I, I]),
    "dpm_corr_kabsch": (I, [P, P, I, LL, P, I, LL, P, P, P, I, I, D, I, D, P, P, P, I, P]),
    "dpm_preprocess_workspace_bytes": (c_size_t, [LL]),
    "dpm_preprocess_scan": (I, [P, I, I, D, D, D, D, LL, P, P, I, P, P, P]),
    "dpm_knn_self_workspace_bytes": (c_size_t, [I]),
    "dpm_knn_self": (I, [P, I, I, D, P, P, P, P, P]),
    "dpm_point_normals": (I, [P, I, D, P, P, P]),
    "dpm_lowpass_similarity": (I, [P, P, I, I, I, P, P]),
    "dpm_stat_filter": (I, [P, I, D, I, D, P, P, P, P, P, P]),
    "dpm_map_tile": (I, [P, P, P, P, I, I, I, P, P]),
    "dpm_infomat_workspace_bytes": (c_size_t, [I, I, I]),
    "dpm_information_matrix": (I, [P, I, P, I, P, D, P, P, P]),
    "dpm_information_matrix_batched": (I, [P, I, P, P, I, P, I, D, P, I, P, P]),
    "dpm_infomat_build_grids": (I, [P, I, P, I, D, P, P]),
    "dpm_infomat_search_grids": (I, [P, I, P, P, I, P, I, D, P, I, P, P]),
}


class DpmError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DpmError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `python deeppointmap_amd/csrc/build.py` (there is no CPU fallback)."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so is stale
            fn.restype, fn.argtypes = res, args
        if lib.dpm_version() & VERSION_EXPERIMENT:
            import sys
            print(f"deeppointmap_amd: {LIB_PATH} is an EXPERIMENTAL build (-DDPM_EXPERIMENT): its kernels read DPM_* "
                  "measurement switches from the environment, some of which skip work -- not for results", file=sys.stderr)
        _lib = lib
    return _lib


def experimental() -> bool:
    """True when the loaded library was built with -DDPM_EXPERIMENT (measurement switches compiled in)."""
    return bool(load().dpm_version() & VERSION_EXPERIMENT)


def check(status: int, what: str) -> None:
    """0 ok; <0 -> ValueError (invalid argument / unsupported shape); >0 -> DpmError (HIP error)."""
    if status == 0:
        return
    msg = load().dpm_error_string(status).decode()
    if status < 0:
        raise ValueError(f"{what}: {msg} (status {status})")
    raise DpmError(f"{what}: HIP error {status}: {msg}")
