"""One-off: Encoder / Decoder under configurations other than the shipped one (widths, expansions, neighbour counts,
level counts, up-sampling depths, extra input channels, decoder widths / depths), against the oracle on small frames.
The reference accepts any of these; the kernels specialise on the shipped shapes and must fall back correctly."""
import os, random, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
from oracle import dpm_oracle as O
torch.set_grad_enabled(False)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = random.Random(seed)
g = torch.Generator().manual_seed(seed)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    cfg = default_args()
    L = rng.choice([3, 4, 5])
    top = rng.choice([512, 1024, 2048])
    cfg.encoder.npoint = [max(top >> (2 * i), 8) for i in range(L)]
    cfg.encoder.radius_list = [[0.05 * 2 ** i, 0.1 * 2 ** i] + ([0.1 * 2 ** i] if rng.random() < 0.3 else []) for i in range(L)]
    ks = rng.choice([8, 16, 32])
    cfg.encoder.nsample_list = [[min(ks, cfg.encoder.npoint[i])] * len(cfg.encoder.radius_list[i]) for i in range(L)]
    cfg.encoder.width = rng.choice([8, 16, 24, 32])
    cfg.encoder.expansion = rng.choice([2, 4])
    cfg.encoder.upsample_layers = rng.randint(1, L - 1)
    cfg.encoder.in_channel = rng.choice([3, 3, 4, 6])
    cfg.encoder.out_channel = rng.choice([64, 128])
    cfg.encoder.sample = [{"type": "fps-t3d"}] * L
    w_ = cfg.encoder.width * 2 ** L                       # encoder.py:39-48: the channel count the up-sampler ends with
    for _ in range(cfg.encoder.upsample_layers):
        cfg.decoder.in_channel = max(cfg.encoder.out_channel, w_ // 2)
        w_ //= 2
    cfg.decoder.model_channel = rng.choice([128, 256])
    cfg.decoder.attention_layers = rng.choice([1, 2, 3])
    desc = (f"npoint {cfg.encoder.npoint} K {ks} width {cfg.encoder.width} exp {cfg.encoder.expansion} up {cfg.encoder.upsample_layers} "
            f"in {cfg.encoder.in_channel} out {cfg.encoder.out_channel} | decoder {cfg.decoder.model_channel} x{cfg.decoder.attention_layers}")
    try:
        enc, dec = init_procedural(Encoder(cfg)).to("cuda:0"), init_procedural(Decoder(cfg)).to("cuda:0")
        sde = {k: v.detach().cpu() for k, v in enc.flat().items()}
        sdd = {k: v.detach().cpu() for k, v in dec.flat().items()}
        B, N = 2, rng.randint(3000, 9000)
        pts, pad = synthetic.frames(B, N, start=rng.randint(0, 50))
        if cfg.encoder.in_channel > 3:
            pts = torch.cat([pts, torch.rand(B, cfg.encoder.in_channel - 3, N, generator=g)], 1)
        coor, fea, mask = enc(pts, pad)
        oc, of, _ = O.encoder_forward(sde, cfg, pts, pad)
        e_xyz, e_fea = bool(torch.equal(coor.cpu(), oc)), float((fea.cpu() - of).abs().max())
        want = torch.cat([of, oc * 60.0], 1)
        R, T, conf, rmse = dec.registration_forward(want[0], want[1], num_sample=0.5)
        Ro, To, co, ro = O.registration_forward(sdd, cfg, want[0], want[1], 0.5)
        dT = float((T.cpu() - To).norm())
        # fewer than ~30 inliers: the loop of decoder.py:227-265 ends on a rank-deficient or empty set and the pose is
        # arbitrary (NaN for an empty one) in the reference as well -- only the inlier count is compared then
        ok = e_xyz and e_fea < 1e-3 and conf.numel() == co.numel() and (co.numel() < 30 or dT < 1e-3)
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed}: {desc}: key points equal {e_xyz}, feature err {e_fea:.2e}, dT {dT:.2e}, inliers {conf.numel()} vs {co.numel()}")
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"ERROR seed {seed}: {desc}: {type(e).__name__}: {str(e)[:200]}")
    n += 1
print(f"seed {seed}: {n} configurations, {bad} problems, {time.time() - t0:.0f} s")
