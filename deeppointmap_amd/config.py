"""Hyper-parameters of the hot path (encoder / decoder / registration).

The values are the ones every shipped inference config of the reference uses
(reference: configs/infer/DeepPointMap_B_Main_SemanticKITTI.yaml:32-60, identical in the
KITTI-360 and MulRan files).  `Cfg` is a tiny attribute-dict that behaves like the
reference's EasyDict for the accesses the hot path performs (`args.encoder.npoint`,
`args.encoder.get('norm', 'LN')`, `args.encoder['expansion']`, `args.loss.tau` ...), so a
reference-side `args` object and ours are interchangeable at the `Encoder(args)` /
`Decoder(args)` boundary (reference: network/encoder/encoder.py:11-22,
network/decoder/decoder.py:12-21).
"""
from __future__ import annotations

import copy


class Cfg(dict):
    """dict with attribute access, recursively applied to nested dicts/lists."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, Cfg):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = self._wrap(v)

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


def default_args(**overrides) -> Cfg:
    """The `args` object of the shipped inference configs (network + slam_system subset)."""
    a = Cfg(
        encoder=dict(
            npoint=[4096, 1024, 256, 64, 16],
            radius_list=[[0.05, 0.1], [0.1, 0.2], [0.2, 0.4, 0.4], [0.4, 0.8], [0.8, 1.6]],
            nsample_list=[[32, 32], [32, 32], [32, 32, 32], [32, 32], [16, 16]],
            in_channel=3,
            out_channel=128,
            width=16,
            expansion=4,
            upsample_layers=2,
            sample=[{"type": "fps-t3d"}] * 5,
            norm="LN",
            bias=True,
        ),
        decoder=dict(in_channel=128, model_channel=256, attention_layers=3),
        loss=dict(tau=0.1, eps_offset=2.0),
        slam_system=dict(
            coor_scale=60,
            odometer_candidates_num=1,
            registration_sample_odometer=0.5,
            registration_sample_mapping=0.5,
            registration_sample_loop=0.5,
        ),
    )
    for k, v in overrides.items():
        a[k] = Cfg._wrap(v)
    return a


def reduced_args() -> Cfg:
    """A shrunken encoder (same topology, fewer points) used by fast parity tests and fixtures."""
    a = default_args()
    a.encoder.npoint = [512, 256, 128, 64, 16]
    return a


def reduced_voxel_args() -> Cfg:
    """reduced_args with voxel samplers at the first and third stage (pointnext.py:21,29-32: the one sampler option no
    shipped config selects); the first keeps the 512 fullest of a few thousand voxels (top-k branch), the third finds
    fewer voxels than it may keep (padded level)."""
    a = reduced_args()
    a.encoder.sample = [{"type": "voxel", "size": 0.02, "range": 1.0}, {"type": "fps-t3d"},
                        {"type": "voxel", "size": 0.25, "range": 1.0}, {"type": "fps-t3d"}, {"type": "fps-t3d"}]
    return a
