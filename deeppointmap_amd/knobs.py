"""Measurement knobs of the Python host side.

The values below ARE the shipped behaviour and no call of the hot path reads the environment.  Scripts under scripts/
(A/B runs) set the attributes directly or call `apply_env()` once at start-up; `bench.py` refuses to run with any `DPM_*`
variable in the environment unless `--allow-knobs` is given, and then records what was applied in its JSON line.  The
kernels' own measurement switches exist only in libraries built with -DDPM_EXPERIMENT (csrc/dpm_common.h: `dpm_knob`),
which `_lib.load()` reports through `_lib.experimental()`.
"""
from __future__ import annotations

import os

FPS_ALGO = None        # int: first-level sampling kernel for 16 384 < N <= 65 536 (None = the library's choice, algo 5)
FUSED_LN = True        # False: Linear + LayerNorm as two kernels at every size (scripts/gemm_ln_shapes.py)
# Linear / Conv1d(k=1) layers with K <= BF16X3_MAX_K on the bf16 matrix pipe as exact three-way splits (csrc/gemm_b3.hip): 18-27 % faster
# alone, +2.7 % frames/s in the pipelined bench, error against fp64 at the fp32 kernel's level.  False: the fp32-MFMA kernel
# everywhere.  (The library is compiled without packed fp32 instructions because of this kernel: csrc/build.py says why.)
GEMM_BF16X3 = True
# layers up to this reduction length take the bf16x3 kernel.  Round 5: 512 -> 2048 (every layer of both networks; the longer ones are
# the encoder's few-row tails 768 -> 256, 1024 -> 256, 2048 -> 512, until then on the fp32-MFMA kernel): pipelined step 4.19 / 4.20 ->
# 4.12 / 4.15 ms (A/B, 60 steps, two alternating repetitions; 1024: 4.14 / 4.12)
BF16X3_MAX_K = 2048
# ... and the Linear + LayerNorm layers with BF16X3_LN_MIN_K <= K <= BF16X3_MAX_K (both forms: fused gemm_ln_b3_kernel / GEMM + LayerNorm, identical
# rows).  Pipelined step 4.27 -> 4.19 ms against the fp32 fused kernel's 4.33 (csrc/gemm_b3.hip, dpm_linear_layernorm_bf16x3,
# has the history of its tile shapes).
GEMM_LN_BF16X3 = True
BF16X3_LN_MIN_K = 128  # shortest reduction a Linear + LayerNorm layer needs to take the bf16x3 kernels (below: the fp32 fused kernel)
# bf16x3 Linear + LayerNorm layers of at most this many rows take the fused kernel as well (identical rows).  Measured for the one-pair
# registration (512 rows): 9 launches fewer, eager call 1.00 -> 0.86 ms, but the replayed graph 0.48 -> 0.61 ms (8 workgroups walking
# K instead of 96): off.
FUSED_LN_SMALL_ROWS = 0
# grouping layers in the folded form (csrc/group_mlp.hip, FOLD): W_r (p - c) / r = W_r' p - W_r' c, the point half added by the
# projection GEMM's epilogue, the centre half one vector per centre.  False: the round-4 form (the relative coordinates formed per
# gathered row).  Moves descriptors by ~3e-6 and poses by ~5e-6 m (cancellation).
FOLD_GATHER = True
# grouping layers with a smaller radius keep the unfolded form (|p| / r is what the fold's rounding error scales with).  0.2: the two
# projected layers of radius 0.1 stay unfolded -- feature error against the oracle over random frames 6.8e-6 median / 1.3e-5 worst
# instead of 1.5e-5 / 1.35e-4 with everything folded, for 1.5 % of the step (round-5 review).  The affine first
# level (radius 0.05) runs a cancellation-free form and is not governed by this.
FOLD_MIN_RADIUS = 0.2
# ... with LayerNorm's mean removal moved into the layer's weights ((I - 11^T / C) W, made once per weight version): the gather
# computes the variance from the rows as they are.  False: the folded form with the mean computed per gathered row.
CENTRED_GATHER = True
FUSED_PWCONV = True    # InvResMLP's pw_conv pair of the first level (C = 32) as one kernel (csrc/gemm_b3.hip, pwconv_pair_b3_kernel); False: two fused GEMM + LayerNorm kernels
FUSED_MATCH = True     # similarity -> dual softmax -> top-k as one operator (csrc/match.hip) where it applies; False: the five-launch form
# the decoder's q | k | v projections hand K and V to the attention kernel as its bf16 operand planes (csrc/gemm_b3.hip KvPlanes,
# csrc/decoder_ops.hip attention_kernel<PRE>): split once by the projection's epilogue instead of by every query block that reads
# a key tile.  Bit-identical to the fp32 hand-over (False).
KV_PLANES = True
DEDUP_FRAMES = True    # False: per-frame decoder work once per pair side instead of once per frame (new Decoder objects)

_ENV = {"DPM_FOLD_MIN_RADIUS": ("FOLD_MIN_RADIUS", float), "DPM_KV_PLANES": ("KV_PLANES", lambda v: v != "0"), "DPM_CENTRED_GATHER": ("CENTRED_GATHER", lambda v: v != "0"), "DPM_FPS_ALGO": ("FPS_ALGO", int), "DPM_NO_FUSED_LN": ("FUSED_LN", lambda v: v != "1"),
        "DPM_DEDUP_FRAMES": ("DEDUP_FRAMES", lambda v: v != "0"), "DPM_GEMM_BF16X3": ("GEMM_BF16X3", lambda v: v == "1"), "DPM_BF16X3_MAX_K": ("BF16X3_MAX_K", int), "DPM_BF16X3_LN_MIN_K": ("BF16X3_LN_MIN_K", int), "DPM_FUSED_LN_SMALL_ROWS": ("FUSED_LN_SMALL_ROWS", int), "DPM_GEMM_LN_BF16X3": ("GEMM_LN_BF16X3", lambda v: v == "1"),
        "DPM_FUSED_MATCH": ("FUSED_MATCH", lambda v: v != "0"), "DPM_FUSED_PWCONV": ("FUSED_PWCONV", lambda v: v != "0"), "DPM_FOLD_GATHER": ("FOLD_GATHER", lambda v: v != "0")}


def env_knobs() -> dict:
    """every DPM_* variable of the environment (host-side knobs, DPM_LIB, and the names an experimental library reads)"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("DPM_")}


def apply_env() -> dict:
    """Set the attributes above from their DPM_* variables, once; returns env_knobs()."""
    g = globals()
    for var, (name, conv) in _ENV.items():
        if var in os.environ:
            g[name] = conv(os.environ[var])
    return env_knobs()
