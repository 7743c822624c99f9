"""Procedural weights, keyed by state-dict name.

The released checkpoint is not available (reference: .MISSING_LARGE_BLOBS:1), so every
parity vector, test and bench run uses weights generated here.  The rule is a pure function
of (key name, shape, seed), so the golden-fixture generator (which loads them into the
*reference* modules with `load_state_dict(strict=True)`), the oracle, the HIP path and the
GPU box all see bit-identical tensors without shipping a checkpoint.

Rule per tensor (numpy RandomState seeded with crc32(key) ^ seed):
  * LayerNorm weight (key ends with `.ln.weight` or `.norm{1,2,3}.weight`): 1 + 0.1*N(0,1)
  * LayerNorm bias:                                                      0.1*N(0,1)
  * other `*.bias` / `in_proj_bias`:                                      0.05*N(0,1)
  * other weights: N(0,1) * gain / sqrt(fan_in), fan_in = shape[1] (conv / linear / in_proj)
"""
from __future__ import annotations

import re
import zlib
from typing import Dict, Mapping, Tuple

import numpy as np
import torch

_LN_RE = re.compile(r"(\.ln|\.norm[123])\.(weight|bias)$")


def _tensor_for(key: str, shape: Tuple[int, ...], seed: int, gain: float) -> np.ndarray:
    rs = np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    m = _LN_RE.search(key)
    x = rs.standard_normal(size=shape)
    if m:
        return (1.0 + 0.1 * x if m.group(2) == "weight" else 0.1 * x).astype(np.float32)
    if key.endswith("bias"):
        return (0.05 * x).astype(np.float32)
    fan_in = shape[1] if len(shape) > 1 else shape[0]
    return (x * (gain / np.sqrt(fan_in))).astype(np.float32)


def procedural_state_dict(shapes: Mapping[str, Tuple[int, ...]], seed: int = 0,
                          gain: float = 1.4142135) -> Dict[str, torch.Tensor]:
    """shapes: {key: shape} (e.g. `{k: tuple(v.shape) for k, v in module.state_dict().items()}`)."""
    return {k: torch.from_numpy(_tensor_for(k, tuple(shapes[k]), seed, gain)) for k in sorted(shapes)}


def init_procedural(module: torch.nn.Module, seed: int = 0, gain: float = 1.4142135) -> torch.nn.Module:
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(procedural_state_dict(shapes, seed, gain), strict=True)
    return module
