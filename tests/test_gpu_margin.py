"""GPU: how far the HIP path is from the EXACT result of the reference's algorithm, next to how far the reference itself is.

tests/golden/margin.npz (tests/golden/make_golden_margin.py, imports the reference) holds, for the twenty registrations the decoder
fuzz found above half the pose tolerance and for the full-size pair the path fuzz found at 7.6e-5 m, the reference's answer three
ways on identical inputs: fp32 as shipped (8 threads), fp32 with one thread (another summation order), and every module and
intermediate in fp64.  The reference's own fp32 answers differ from its fp64 answer by up to 1.3e-4 m in the 'margin' class and
2.5e-4 m on the full-size pair, and from EACH OTHER (thread count) by up to 1.1e-4 m / 0.30 m -- the north_star's 1e-4 m is tighter
than the reference's reproducibility on these inputs.  The statement tested here is the one that can be true of any fp32
implementation: no further from the fp64 result than a small multiple of the reference's own distance.

  'margin'    cases (same pairs, same inliers in every evaluation): |HIP - ref64| <= max(3 |ref32 - ref64|, 1e-4 m / 1e-4 rad), inlier
              count equal to the reference's.
  'boundary'  cases (a discrete decision of the reference sits at rounding level): the HIP pose equals one of the reference's three
              evaluations within 1e-4 m / 1e-4 rad, or the fp64 evaluation itself shows the decision at rounding level (k-th pair
              confidence / 64th seed weight within 2e-3 relative of the next -- confidences agree to ~1e-3 between fp32 and fp64 --,
              offset-outlier cut or inlier cut within 1e-4 relative).
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, rot_angle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import margin_cases  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
C, FLOOR = 3.0, 1e-4


@pytest.fixture(scope="module")
def dec(cfg_full):
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.weights import init_procedural
    return init_procedural(Decoder(cfg_full)).to(DEV)


def _dist(R, T, g, key):
    Rr, Tr = torch.from_numpy(g[key + ".R"]), torch.from_numpy(g[key + ".T"])
    return float((T.double() - Tr).norm()), rot_angle(R.double(), Rr)


def _three(g, k):
    a, b = torch.from_numpy(g[k + ".ref32.T"]), torch.from_numpy(g[k + ".ref64.T"])
    return float((a - b).norm()), rot_angle(torch.from_numpy(g[k + ".ref32.R"]), torch.from_numpy(g[k + ".ref64.R"]))


CASES = margin_cases.all_cases() if torch.cuda.is_available() else []


@pytest.mark.parametrize("case", CASES, ids=[f"s{s}_n{n}_{cls}" for (s, n), cls, _ in CASES])
def test_fuzz_cases_against_the_reference_in_fp64(case, dec):
    (seed, n), cls, (s, d, ms, md, ns) = case
    g = load_golden("margin.npz")
    k = f"s{seed}_n{n}"
    R, T, conf, rmse = dec.registration_forward(s, d, ms, md, num_sample=ns)
    R, T = R.cpu(), T.cpu()
    dT64, dR64 = _dist(R, T, g, k + ".ref64")
    eT, eR = _three(g, k)            # the reference's own fp32 distance from its fp64 result
    near64 = dT64 <= max(C * eT, FLOOR) and dR64 <= max(C * eR, FLOOR)
    if cls == "margin":
        assert int(conf.numel()) == int(g[k + ".ref32.n_conf"]) == int(g[k + ".ref64.n_conf"])
        assert near64, f"{k}: HIP {dT64:.2e} m / {dR64:.2e} rad from the fp64 result, the reference's fp32 {eT:.2e} m / {eR:.2e} rad"
        return
    same_branch = any(max(_dist(R, T, g, f"{k}.{tag}")[0], _dist(R, T, g, f"{k}.{tag}")[1]) <= FLOOR for tag in ("ref32", "ref32t1", "ref64"))
    m = {q: float(g[f"{k}.ref64.margin_{q}"]) for q in ("kgap", "sgap", "ocut", "inlier")}
    at_rounding_level = min(m["kgap"], m["sgap"]) < 2e-3 or min(m["ocut"], m["inlier"]) < 1e-4
    assert near64 or same_branch or at_rounding_level, (k, dT64, dR64, m)
    assert at_rounding_level, f"{k} is listed as a boundary case but the fp64 evaluation shows no decision near its cut: {m}"


@pytest.mark.parametrize("k", ["path701_702", "path900_901"])
def test_full_size_pair_of_the_path_fuzz_against_the_reference_in_fp64(k, dec):
    """synthetic frames 701 -> 702 and 900 -> 901 at 65 536 points (scripts/fuzz_path.py's worst pairs: 7.6e-5 m from the oracle in round
    5, 1.04e-4 m in round 6): the decoder on the REFERENCE's fp32 descriptors of the two frames.  The reference's own three evaluations
    of 701 -> 702: fp32 vs fp64 2.5e-4 m, 8 threads vs 1 thread 1.1e-4 m, 85 inliers each (900 -> 901: the generator's table)."""
    g = load_golden("margin.npz")
    s, d = torch.from_numpy(g[k + ".desc_src"]), torch.from_numpy(g[k + ".desc_dst"])
    R, T, conf, rmse = dec.registration_forward(s, d, num_sample=0.5)
    R, T = R.cpu(), T.cpu()
    assert int(conf.numel()) == int(g[k + ".ref32.n_conf"]) == int(g[k + ".ref64.n_conf"])
    dT64, dR64 = _dist(R, T, g, k + ".ref64")
    eT, eR = _three(g, k)
    assert dT64 <= max(C * eT, FLOOR) and dR64 <= max(C * eR, FLOOR), (dT64, dR64, eT, eR)
    # and it stays where the fixtures' tolerance puts it with respect to the reference as shipped
    dT32, dR32 = _dist(R, T, g, k + ".ref32")
    spread = float(np.linalg.norm(g[k + ".ref32.T"] - g[k + ".ref32t1.T"]))   # the reference against itself
    assert dT32 <= max(FLOOR, 1.5 * spread) and dR32 <= FLOOR, (dT32, dR32, spread)


@pytest.mark.parametrize("fold_min_radius", [None, 0.0], ids=["shipped", "folded-at-every-radius"])
def test_descriptors_against_the_reference_in_fp64(fold_min_radius, cfg_full, monkeypatch):
    """tests/golden/encoder_noise.npz: frames 0 and 1 of the synthetic sequence (65 536 points) through the reference encoder in fp64
    (key points identical to the fp32 run).  The reference's own fp32 descriptors sit 8e-5 (90th percentile) / 8.7e-4 (max) from that
    result; the HIP path's are held to the same distance plus the feature tolerance -- in the shipped configuration, and with the
    grouping layers folded at EVERY radius (knobs.FOLD_MIN_RADIUS = 0: the form round 5 shipped, 1.5 % faster, feature error against
    the fp32 oracle 1.5e-5 median / 1.35e-4 worst over random frames): the price of the full fold is inside the reference's own
    distance from the exact result, which is the evidence a deployment that wants the 1.5 % needs."""
    from deeppointmap_amd import knobs, synthetic
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.weights import init_procedural
    from conftest import FEATURE_TOL
    if fold_min_radius is not None:
        monkeypatch.setattr(knobs, "FOLD_MIN_RADIUS", fold_min_radius)
    g, ref = load_golden("encoder_noise.npz"), load_golden("encoder_full.npz")
    enc = init_procedural(Encoder(cfg_full)).to(DEV)
    for f in (0, 1):
        p = synthetic.frame(f).unsqueeze(0).to(DEV)
        coor, fea, _ = enc(p, torch.zeros(1, p.shape[2], dtype=torch.bool, device=DEV))
        assert bool(g[f"synthetic{f}.coor64_equal"])
        assert np.array_equal(coor[0].cpu().numpy().astype(np.float64), g[f"synthetic{f}.coor64"])      # key points: exact, also against fp64
        hip, f64, f32 = fea[0].cpu().numpy().astype(np.float64), g[f"synthetic{f}.fea64"], ref[f"synthetic{f}.fea"].astype(np.float64)
        own = np.abs(f32 - f64)                     # the reference's fp32 distance from its fp64 result
        ours = np.abs(hip - f64)
        scale = max(1.0, float(np.abs(f64).max()))
        assert ours.max() <= own.max() + FEATURE_TOL * scale, (f, ours.max(), own.max())
        assert np.percentile(ours, 90) <= np.percentile(own, 90) + FEATURE_TOL * scale / 3, (f, np.percentile(ours, 90), np.percentile(own, 90))
        print(f"frame {f} ({'shipped' if fold_min_radius is None else 'fold everywhere'}): |HIP - ref64| max {ours.max():.2e} p90 {np.percentile(ours, 90):.2e}; "
              f"|ref32 - ref64| max {own.max():.2e} p90 {np.percentile(own, 90):.2e}; |HIP - ref32| max {np.abs(hip - f32).max():.2e}")
