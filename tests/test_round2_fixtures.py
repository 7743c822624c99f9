"""Round-2 reference fixtures (tests/golden/make_golden_r2.py): tie-heavy small-N neighbour queries, a converged
registration, and the map-sized registrations of BASELINE configs 3-5.  Inputs come from tests/golden/r2_cases.py."""
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, T, assert_features_close, idx_rows_equal_as_sets, load_golden, rot_angle
from oracle import dpm_oracle as O

sys.path.insert(0, GOLDEN)
import r2_cases  # noqa: E402

DEV = "cuda:0"
TOL_T, TOL_R = 1e-4, 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# kNN with exactly tied distances across the K-th place, k * 64 > n: torch.topk's std::nth_element branch
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_knn_lattice_ties_vs_reference():
    g = load_golden("knn_ties.npz")
    for name, (pts, ctr, r, K) in r2_cases.knn_tie_cases().items():
        idx = O.hybrid_query(r, K, pts.unsqueeze(0), ctr.unsqueeze(0), torch.zeros(1, pts.shape[0], dtype=torch.bool))[0]
        assert np.array_equal(idx.numpy().astype(np.int32), g[name + ".idx"]), name


@pytest.mark.gpu
def test_hip_knn_lattice_ties_vs_reference():
    from deeppointmap_amd import ops
    g = load_golden("knn_ties.npz")
    for name, (pts, ctr, r, K) in r2_cases.knn_tie_cases().items():
        lens = torch.tensor([pts.shape[0]], dtype=torch.int32, device=DEV)
        idx = ops.knn_hybrid(pts.unsqueeze(0).to(DEV), lens, ctr.unsqueeze(0).to(DEV), K, r)[0].cpu().numpy()
        want = g[name + ".idx"]
        # the reference's rows as SETS, every row: with ties across the K-th place this pins WHICH of the equally distant
        # points the reference keeps; and slot 0 is the nearest point (the centre itself)
        same = idx_rows_equal_as_sets(idx, want)
        assert same.all(), (name, int((~same).sum()), idx[~same][:2], want[~same][:2])
        assert (idx[:, 0] == want[:, 0]).all(), name


# ---------------------------------------------------------------------------------------------------------------------
# a registration that converges: the `conf > 0.5` inliers, rmse below the SLAM system's 0.5 m gate (mapping.py:52-81)
# ---------------------------------------------------------------------------------------------------------------------
def _converged_cfg():
    from deeppointmap_amd.config import default_args
    cfg = default_args()
    cfg.loss.tau = r2_cases.CONVERGED_TAU
    return cfg


def test_oracle_converged_registration_vs_reference(sd_dec):
    g = load_golden("converged.npz")
    cfg, sd = _converged_cfg(), r2_cases.converged_state_dict(sd_dec)
    for name, (src, dst, Rgt, tgt) in r2_cases.converged_cases().items():
        R, T_, conf, rmse = O.registration_forward(sd, cfg, src, dst, 0.5)
        assert float(g[name + ".rmse"]) < 0.5 and int(g[name + ".n_high"]) > 64  # the regime the fixture is for
        assert float((T_ - T(g[name + ".T"])).norm()) < TOL_T and rot_angle(R, g[name + ".R"]) < TOL_R, name
        assert conf.numel() == int(g[name + ".n_conf"]) and int((conf > 0.5).sum()) == int(g[name + ".n_high"])
        assert abs(rmse - float(g[name + ".rmse"])) < 1e-5


@pytest.mark.gpu
def test_hip_converged_registration_vs_reference(sd_dec):
    from deeppointmap_amd.decoder import Decoder
    g = load_golden("converged.npz")
    dec = Decoder(_converged_cfg())
    dec.load_state_dict(r2_cases.converged_state_dict(sd_dec), strict=True)
    dec = dec.to(DEV)
    for name, (src, dst, Rgt, tgt) in r2_cases.converged_cases().items():
        tr = {}
        R, T_, conf, rmse = dec.registration_forward(src, dst, num_sample=0.5, trace=tr)
        np.testing.assert_allclose(tr["conf"].cpu().numpy().reshape(-1), g[name + ".pair_conf"], rtol=2e-3, atol=1e-6)
        dT, dR = float((T_.cpu() - T(g[name + ".T"])).norm()), rot_angle(R.cpu(), g[name + ".R"])
        assert dT < TOL_T and dR < TOL_R, (name, dT, dR)
        assert conf.numel() == int(g[name + ".n_conf"]) and int((conf > 0.5).sum()) == int(g[name + ".n_high"])
        assert abs(rmse - float(g[name + ".rmse"])) < 1e-5 and rmse < 0.5
        assert float((T_.cpu() - tgt).norm()) < 0.15  # and it is the motion the target was built with


# ---------------------------------------------------------------------------------------------------------------------
# map-sized registrations: tiles from the reference's PoseGraph.global_map_query_graph, 4096 x 256 and 4096 x 4096
# ---------------------------------------------------------------------------------------------------------------------
def _tiles_oracle(g):
    kps, poses = r2_cases.keyframe_pool()
    tiles = {}
    for n, c in (("a", r2_cases.TILE_A), ("b", r2_cases.TILE_B)):
        toks = [int(t) for t in g[f"tile_{n}.tokens"]]
        tiles[n] = O.map_tile([kps[t] for t in toks], [poses[t] for t in toks], poses[c])
        np.testing.assert_allclose(tiles[n][128:].numpy(), g[f"tile_{n}.xyz"], atol=3e-5)
    return kps, poses, tiles


def test_oracle_scan_to_map_4096x256_vs_reference(cfg_full, sd_dec):
    g = load_golden("large_reg.npz")
    kps, poses, tiles = _tiles_oracle(g)
    assert tiles["a"].shape == (131, 4096) and len(set(g["tile_a.tokens"].tolist()) & set(g["tile_b.tokens"].tolist())) == 0
    R, T_, conf, rmse = O.registration_forward(sd_dec, cfg_full, tiles["a"], kps[r2_cases.SCAN], 0.5)
    assert g["s2m_4096x256.pair_conf"].size == 1088
    assert float((T_ - T(g["s2m_4096x256.T"])).norm()) < TOL_T and rot_angle(R, g["s2m_4096x256.R"]) < TOL_R
    assert conf.numel() == int(g["s2m_4096x256.n_conf"])


@pytest.mark.gpu
def test_hip_map_sized_registrations_vs_reference(cfg_full):
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.maptile import MapTileStore
    from deeppointmap_amd.registration import simvec_to_num
    from deeppointmap_amd.weights import init_procedural
    g = load_golden("large_reg.npz")
    kps, poses = r2_cases.keyframe_pool()
    dec = init_procedural(Decoder(cfg_full)).to(DEV)
    store = MapTileStore(DEV)
    for t, kp in enumerate(kps):
        store.put(t, kp)
    tiles = {}
    for n, c in (("a", r2_cases.TILE_A), ("b", r2_cases.TILE_B)):
        toks = [int(t) for t in g[f"tile_{n}.tokens"]]
        tiles[n], _ = store.tile(toks, [poses[t] for t in toks], poses[c])
        assert tuple(tiles[n].shape) == (131, 4096)
        np.testing.assert_allclose(tiles[n][128:].cpu().numpy(), g[f"tile_{n}.xyz"], atol=3e-5)
    for name, (s, d, k) in {"s2m_4096x256": (tiles["a"], kps[r2_cases.SCAN], 1088),
                            "loop_4096x4096": (tiles["a"], tiles["b"], 2048)}.items():
        tr = {}
        R, T_, conf, rmse = dec.registration_forward(s, d, num_sample=0.5, trace=tr)
        assert tr["conf"].numel() == k == g[name + ".pair_conf"].size
        np.testing.assert_allclose(tr["conf"].cpu().numpy().reshape(-1), g[name + ".pair_conf"], rtol=3e-3, atol=1e-9)
        assert tr["n_corr"] == int(g[name + ".n_corr"])
        dT, dR = float((T_.cpu() - T(g[name + ".T"])).norm()), rot_angle(R.cpu(), g[name + ".R"])
        assert dT < TOL_T and dR < TOL_R, (name, dT, dR)
        assert conf.numel() == int(g[name + ".n_conf"]) and abs(rmse - float(g[name + ".rmse"])) < 1e-4
        assert abs(simvec_to_num(conf) - float(g[name + ".conf30"])) < 1e-5
    S = torch.stack([kps[i] for i in r2_cases.LOOP_SRC])
    D = kps[r2_cases.LOOP_DST].unsqueeze(0).repeat(S.shape[0], 1, 1)
    assert S.shape[0] == 16
    np.testing.assert_allclose(dec.loop_detection_forward(S, D).cpu().numpy(), g["loop16.prob"], atol=2e-5)


# ---------------------------------------------------------------------------------------------------------------------
# padding masks: nn.MultiheadAttention's key_padding_mask in every attention block (descriptor_attention.py:33-42)
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_padding_masks_vs_reference(cfg_full, sd_dec):
    g = load_golden("masked.npz")
    for name, (src, dst, ms, md) in r2_cases.masked_cases().items():
        if name.startswith("reg"):
            R, T_, conf, rmse = O.registration_forward(sd_dec, cfg_full, src[0], dst[0], 0.5, src_padding_mask=ms, dst_padding_mask=md)
            assert float((T_ - T(g[name + ".T"])).norm()) < TOL_T and rot_angle(R, g[name + ".R"]) < TOL_R, name
            assert conf.numel() == int(g[name + ".n_conf"]) and abs(rmse - float(g[name + ".rmse"])) < 1e-4
        else:
            prob = O.loop_detection_forward(sd_dec, cfg_full, src, dst, ms, md)
            np.testing.assert_allclose(prob.numpy(), g[name + ".prob"], atol=2e-6)
    src, dst, ms, md = r2_cases.masked_cases()["reg_256"]
    x, _, y, _ = O.descriptor_attention(sd_dec, cfg_full, src, dst, ms, md)
    np.testing.assert_allclose(x[0].t().numpy(), g["reg_256.src_corr"], atol=2e-4)
    # and the masks matter: without them the same inputs give a different answer
    x0, _, _, _ = O.descriptor_attention(sd_dec, cfg_full, src, dst)
    assert float((x0 - x).abs().max()) > 1e-2


@pytest.mark.gpu
def test_hip_padding_masks_vs_reference(cfg_full):
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.registration import simvec_to_num
    from deeppointmap_amd.weights import init_procedural
    g = load_golden("masked.npz")
    dec = init_procedural(Decoder(cfg_full)).to(DEV)
    for name, (src, dst, ms, md) in r2_cases.masked_cases().items():
        if name.startswith("reg"):
            tr = {}
            R, T_, conf, rmse = dec.registration_forward(src[0], dst[0], ms, md, num_sample=0.5, trace=tr)
            if name + ".src_corr" in g:
                M, N = src.shape[2], dst.shape[2]
                assert_features_close(tr["x"].view(M, -1).t().cpu().numpy(), g[name + ".src_corr"], f"masked {name} correlated src features")
                assert_features_close(tr["y"].view(N, -1).t().cpu().numpy(), g[name + ".dst_corr"], f"masked {name} correlated dst features")
            dT, dR = float((T_.cpu() - T(g[name + ".T"])).norm()), rot_angle(R.cpu(), g[name + ".R"])
            assert dT < TOL_T and dR < TOL_R, (name, dT, dR)
            assert conf.numel() == int(g[name + ".n_conf"]) and abs(rmse - float(g[name + ".rmse"])) < 1e-4
            assert abs(simvec_to_num(conf) - float(g[name + ".conf30"])) < 1e-5
        else:
            prob = dec.loop_detection_forward(src, dst, ms, md)
            np.testing.assert_allclose(prob.cpu().numpy(), g[name + ".prob"], atol=2e-5)
    # a mask of the wrong shape is an error, not a silent broadcast
    src, dst, ms, md = r2_cases.masked_cases()["reg_256"]
    with pytest.raises(ValueError):
        dec.registration_forward(src[0], dst[0], ms[:, :100], md)


# ---------------------------------------------------------------------------------------------------------------------
# Sampler('fps')(random_start_point=True): the start index is random.randint(0, length - 1), one draw per frame
# ---------------------------------------------------------------------------------------------------------------------
def test_oracle_fps_random_start_vs_reference():
    import random
    g = load_golden("fps_random_start.npz")
    random.seed(r2_cases.RANDOM_START_SEED)
    for name, (pts, pad, K) in r2_cases.random_start_cases().items():
        lengths = (~pad).sum(1)
        for b in range(pts.shape[0]):
            idx = O.fps_indices(pts[b], int(lengths[b]), K, start=random.randint(0, int(lengths[b]) - 1))
            assert np.array_equal(O.gather_masked(pts[b], idx).numpy(), g[name + ".points"][b]), (name, b)
            assert np.array_equal((idx < 0).numpy(), g[name + ".mask"][b])


@pytest.mark.gpu
def test_hip_fps_random_start_vs_reference():
    import random
    from deeppointmap_amd.operators import Sampler
    g = load_golden("fps_random_start.npz")
    fps = Sampler("fps-t3d")
    random.seed(r2_cases.RANDOM_START_SEED)
    for name, (pts, pad, K) in r2_cases.random_start_cases().items():
        new, mask = fps(points=pts.to(DEV), points_padding=pad.to(DEV), K=K, random_start_point=True)
        assert np.array_equal(new.cpu().numpy(), g[name + ".points"]), name   # same draws, same picks, same order
        assert np.array_equal(mask.cpu().numpy(), g[name + ".mask"]), name
    # and through every algorithm that takes a start index, at the size the Sort-Tile-Recursive packing serves
    from deeppointmap_amd import ops
    gen = torch.Generator().manual_seed(3)
    xyz = (torch.randn(2, 30000, 3, generator=gen) * torch.tensor([40.0, 40.0, 2.0])).to(DEV)
    lens = torch.tensor([30000, 17000], dtype=torch.int32, device=DEV)
    start = torch.tensor([12345, 16999], dtype=torch.int32, device=DEV)
    got = ops.fps(xyz, lens, 300, start=start)[0].cpu()
    for b in range(2):
        want = O.fps_indices(xyz[b].cpu(), int(lens[b]), 300, start=int(start[b]))
        assert torch.equal(got[b].long(), want), b
