"""State-dict layout of the reference's Encoder / Decoder (key names and shapes).

The checkpoint format is part of the drop-in contract: pipeline/infer.py:63-65 calls
`encoder.load_state_dict(weights['encoder'], strict=True)` and
`decoder.load_state_dict(weights['decoder'], strict=False)`.  The key sets below are derived
from the reference constructors (network/encoder/encoder.py:24-49, pointnext.py:28,81,123-126,
153-166,186; network/decoder/decoder.py:24-32, descriptor_attention.py:12-22, heads.py:6-62)
and verified by loading them with strict=True into the reference modules
(tests/golden/make_golden.py).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import torch
import torch.nn as nn

Shapes = "OrderedDict[str, Tuple[int, ...]]"


def _conv(out: Shapes, key: str, cout: int, cin: int, dim: int):
    out[key + ".weight"] = (cout, cin) + (1,) * dim
    out[key + ".bias"] = (cout,)


def _ln(out: Shapes, key: str, c: int):
    out[key + ".weight"] = (c,)
    out[key + ".bias"] = (c,)


def encoder_shapes(cfg) -> Shapes:
    enc = cfg.encoder
    if str(enc.get("norm", "LN")).lower() != "ln" or not enc.get("bias", True):
        raise NotImplementedError("only norm=LN, bias=True (every shipped config) is implemented")
    s: Shapes = OrderedDict()
    width = enc.width
    _conv(s, "point_mlp0", width, enc.in_channel, 1)
    for i in range(len(enc.npoint)):
        c, p = width, f"downsampler.{i}"
        _conv(s, p + ".sa.mlp.0", 2 * c, c + 3, 2)
        _ln(s, p + ".sa.mlp.1.ln", 2 * c)
        c2 = 2 * c
        for j in range(len(enc.radius_list[i]) - 1):
            q = f"{p}.irm.{j}"
            _conv(s, q + ".la.mlp.0", c2, c2 + 3, 2)
            _ln(s, q + ".la.mlp.1.ln", c2)
            _conv(s, q + ".pw_conv.0", c2 * enc["expansion"], c2, 1)
            _ln(s, q + ".pw_conv.1.ln", c2 * enc["expansion"])
            _conv(s, q + ".pw_conv.3", c2, c2 * enc["expansion"], 1)
            _ln(s, q + ".pw_conv.4.ln", c2)
        width *= 2
    up_in = width
    for i in range(enc.upsample_layers):
        up_out = max(enc.out_channel, width // 2)
        p = f"upsampler.{i}"
        _conv(s, p + ".mlp.0", up_out, up_in + width // 2, 1)
        _ln(s, p + ".mlp.1.ln", up_out)
        _conv(s, p + ".mlp.3", up_out, up_out, 1)
        _ln(s, p + ".mlp.4.ln", up_out)
        width //= 2
        up_in = up_out
    return s


def decoder_shapes(cfg) -> Shapes:
    dec = cfg.decoder
    E, Cin = dec.model_channel, dec.in_channel
    s: Shapes = OrderedDict()
    _conv(s, "projection", E, Cin, 1)
    for l in range(dec.attention_layers):
        p = f"descriptor_attention.{l}"
        for a in ("self_attn", "cross_attn"):
            s[f"{p}.{a}.in_proj_weight"] = (3 * E, E)
            s[f"{p}.{a}.in_proj_bias"] = (3 * E,)
            _conv(s, f"{p}.{a}.out_proj", E, E, 0)
        _conv(s, p + ".mlp.0", E, E, 0)
        _conv(s, p + ".mlp.2", E, E, 0)
        for n in ("norm1", "norm2", "norm3"):
            _ln(s, f"{p}.{n}", E)
    _conv(s, "similarity_head.0", E, E, 1)
    _conv(s, "similarity_head.2", E, E, 1)
    E2 = 2 * E
    _conv(s, "offset_head.mlp.0", E2 // 2, E2, 1)
    _conv(s, "offset_head.mlp.2", E2 // 4, E2 // 2, 1)
    _conv(s, "offset_head.mlp.4", E2 // 8, E2 // 4, 1)
    _conv(s, "offset_head.downsample", E2 // 8, E2, 1)
    _conv(s, "offset_head.head", 3, E2 // 8, 1)
    _conv(s, "loop_head.mlp.0", E, E, 1)
    _conv(s, "loop_head.mlp.2", E, E, 1)
    _conv(s, "loop_head.projection.0", E2, E2, 0)
    _conv(s, "loop_head.projection.2", 1, E2, 0)
    _conv(s, "coarse_pairing_head.0", Cin, Cin, 1)  # training-only head; kept so checkpoints load
    _conv(s, "coarse_pairing_head.2", Cin, Cin, 1)
    return s


class ParamTree(nn.Module):
    """nn.Module whose parameters carry the given dotted names (nested holder modules)."""

    def __init__(self, shapes: Shapes = None):
        super().__init__()
        self._flat: Dict[str, nn.Parameter] = {}
        self._epoch = 0   # bumped whenever parameter storage may have moved (load_state_dict, .to(), invalidate_caches)
        for key, shape in (shapes or {}).items():
            self._add(key, shape)

    def _add(self, key: str, shape):
        node, parts = self, key.split(".")
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, _Node())
            node = node._modules[part]
        p = nn.Parameter(torch.zeros(shape, dtype=torch.float32), requires_grad=False)
        node.register_parameter(parts[-1], p)
        self._flat[key] = p

    def p(self, key: str) -> torch.Tensor:
        return self._flat[key]

    def invalidate_caches(self) -> None:
        """Drop the tensors the kernels derived from these weights (folded point_mlp0, packed feature columns ...).
        Automatic on load_state_dict(); needed by hand only after in-place edits through `.data`."""
        from . import ops
        ops.invalidate_derived()
        self._epoch += 1

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda() / .float(): the parameters' storage moves"""
        out = super()._apply(fn, *args, **kwargs)
        self._epoch += 1
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_caches()
        return out

    def flat(self) -> Dict[str, torch.Tensor]:
        """{key: tensor} view of the live parameters (what the oracle calls `sd`)."""
        return dict(self._flat)


class _Node(nn.Module):
    pass
