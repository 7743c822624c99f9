"""GPU: Encoder.forward (HIP path) against reference fixtures and the oracle."""
import numpy as np
import pytest
import torch

from conftest import T, assert_features_close, idx_rows_equal_as_sets, load_golden
from oracle import dpm_oracle as O
from deeppointmap_amd import synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_encoder(cfg):
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.weights import init_procedural
    return init_procedural(Encoder(cfg)).to(DEV)


@pytest.mark.parametrize("fixture", ["encoder_reduced.npz", "encoder_reduced_padded.npz", "encoder_reduced_voxel.npz"])
def test_encoder_reduced_per_stage_vs_reference(fixture, cfg_reduced):
    g = load_golden(fixture)
    if "lengths" in g:   # voxel samplers at stages 0 and 2 (config.reduced_voxel_args), second frame ragged
        from deeppointmap_amd.config import reduced_voxel_args
        cfg_reduced = reduced_voxel_args()
    enc = make_encoder(cfg_reduced)
    pts = T(g["points"])
    if "lengths" in g:
        pad = torch.arange(pts.shape[2]).unsqueeze(0) >= T(g["lengths"]).unsqueeze(1)
    elif "length" in g:
        pad = torch.arange(pts.shape[2]).unsqueeze(0) >= int(g["length"])
    else:
        pad = torch.zeros(pts.shape[0], pts.shape[2], dtype=torch.bool)
    tr = {}
    coor, fea, mask = enc(pts, pad, trace=tr)  # CPU inputs: the module stages them
    assert coor.is_cuda and fea.is_cuda
    assert np.array_equal(coor.cpu().numpy(), g["coor"]), "keypoint coordinates must be bit-identical"
    assert np.array_equal(mask.cpu().numpy(), g["mask"])
    for k, v in g.items():
        if k.endswith(".fps.new"):
            assert np.array_equal(tr[k].cpu().numpy(), v), k
        elif k.endswith(".idx") and not k.endswith("fps.idx"):
            got = tr[k].cpu().numpy().reshape(v.shape)
            # every row, as a set: ties at the K-th place are resolved the way torch.topk resolves them (topk_emulate.h)
            assert idx_rows_equal_as_sets(got.reshape(-1, v.shape[-1]), v.reshape(-1, v.shape[-1])).all(), k
        elif k.endswith(".out"):
            assert_features_close(tr[k].cpu().numpy(), v, f"{fixture} {k}")
    assert_features_close(fea.cpu().numpy(), g["fea"], f"{fixture} descriptors")


@pytest.mark.parametrize("tag", ["synthetic0", "synthetic1", "kitti0", "kitti1"])
def test_encoder_full_descriptors_vs_reference(tag, cfg_full):
    g = load_golden("encoder_full.npz")
    enc = make_encoder(cfg_full)
    if tag.startswith("synthetic"):
        p = synthetic.frame(int(tag[-1])).unsqueeze(0)
    else:
        p = T(g[tag + ".points"]).unsqueeze(0)
    coor, fea, mask = enc(p.to(DEV), torch.zeros(1, p.shape[2], dtype=torch.bool, device=DEV))
    assert tuple(coor.shape) == (1, 3, 256) and tuple(fea.shape) == (1, 128, 256) and not bool(mask.any())
    assert np.array_equal(coor[0].cpu().numpy(), g[tag + ".coor"])
    assert_features_close(fea[0].cpu().numpy(), g[tag + ".fea"], f"encoder_full {tag} descriptors")


def test_encoder_batch_equals_single_frames(cfg_full):
    enc = make_encoder(cfg_full)
    pts, pad = synthetic.frames(3, 16384)
    coor, fea, _ = enc(pts, pad)
    for b in range(3):
        c1, f1, _ = enc(pts[b:b + 1], pad[b:b + 1])
        assert torch.equal(c1[0], coor[b]) and torch.equal(f1[0], fea[b])


def test_encoder_rejects_cpu_module(cfg_reduced):
    from deeppointmap_amd.encoder import Encoder
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Encoder(cfg_reduced)(torch.zeros(1, 3, 64), torch.zeros(1, 64, dtype=torch.bool))


def test_nested_fps_prefix_equals_explicit_chain(cfg_full):
    """Levels 1.. of the sampling chain are taken as prefixes of level 0 (Encoder.nested_fps); running every level
    explicitly must give the very same tensors -- on regular scans, on a lattice (every distance tied), with fewer
    distinct points than picks at the lower levels, and on ragged frames."""
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.weights import init_procedural
    enc = init_procedural(Encoder(cfg_full)).to(DEV)
    gen = torch.Generator().manual_seed(31)
    N = 20000
    pts = torch.rand(4, 3, N, generator=gen)
    g = torch.stack(torch.meshgrid(torch.arange(28.0), torch.arange(28.0), torch.arange(28.0), indexing="ij"), -1).reshape(-1, 3)
    pts[1] = (g[torch.randperm(g.shape[0], generator=gen)][:N] * 0.03).t()
    pts[2, :, 2000:] = pts[2, :, torch.randint(0, 2000, (N - 2000,), generator=gen)]   # 2000 distinct points only
    pad = torch.arange(N).unsqueeze(0) >= torch.tensor([[N], [N], [N], [3000]])
    assert enc.nested_fps
    a = enc.presample(pts, pad)
    enc.nested_fps = False
    b = enc.presample(pts, pad)
    assert a.keys() == b.keys()
    for k in a:
        if k.startswith("fidx") and k != "fidx0":
            # a frame with fewer distinct points than picks repeats its first point; the explicit chain then reports
            # position 0 for the repeats, the prefix their own position -- same coordinates either way
            same = (a[k] == b[k]) | (b[k] == 0)
            assert bool(same.all()), k
        elif k == "grids":  # the pre-built search grids: same set of queries (their contents are covered by the kNN tests)
            assert a[k].keys() == b[k].keys()
        else:
            assert torch.equal(a[k], b[k]), k


def test_neighbour_queries_in_the_geometry_pass_change_nothing(cfg_full):
    """Encoder.presample_neighbours moves the neighbour queries into presample() (a pipeline balancing knob): the
    descriptors are the same bits -- the max over the K slots does not see their order."""
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.weights import init_procedural
    enc = init_procedural(Encoder(cfg_full)).to(DEV)
    pts, pad = synthetic.frames(2, 30000)
    pad[1, 21000:] = True
    coor0, fea0, mask0 = enc(pts, pad)
    enc.presample_neighbours = True
    pre = enc.presample(pts, pad)
    assert len(pre["knn"]) >= 6
    coor1, fea1, mask1 = enc(pts, pad, presampled=pre)
    assert torch.equal(coor0, coor1) and torch.equal(fea0, fea1) and torch.equal(mask0, mask1)


def test_encoder_extra_input_channels_vs_oracle(cfg_reduced):
    """in_channel > 3 (e.g. intensity as a fourth point channel; no shipped config uses it): point_mlp0 is then a
    real GEMM over the input channels and the first set abstraction reads materialised features.  Checked against
    the oracle (the reference module is identical up to its state-dict shapes, which the procedural weights fill)."""
    import copy
    cfg = copy.deepcopy(cfg_reduced)
    cfg.encoder.in_channel = 4
    enc = make_encoder(cfg)
    sd = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
    assert tuple(sd["point_mlp0.weight"].shape[:2]) == (cfg.encoder.width, 4)
    gen = torch.Generator().manual_seed(41)
    pts = torch.cat([synthetic.frames(2, 4096)[0], torch.rand(2, 1, 4096, generator=gen)], dim=1)   # x, y, z, intensity
    pad = torch.zeros(2, 4096, dtype=torch.bool)
    coor, fea, mask = enc(pts, pad)
    wc, wf, wm = O.encoder_forward(sd, cfg, pts, pad)
    assert torch.equal(coor.cpu(), wc) and torch.equal(mask.cpu(), wm)
    assert_features_close(fea.cpu().numpy(), wf.numpy(), "in_channel=4 descriptors vs oracle")
    # the extra channel really reaches the descriptors
    pts2 = pts.clone()
    pts2[:, 3] = 1.0 - pts2[:, 3]
    assert float((enc(pts2, pad)[1] - fea).abs().max()) > 1e-3
