"""One-off: the whole path (encode -> register -> information matrix) at full size on frames far along the synthetic
trajectory (what ranks 1..7 of a sharded run see), against the oracle: poses within 1e-4 m / 1e-4 rad."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural
from oracle import dpm_oracle as O
torch.set_grad_enabled(False)
cfg = default_args()
enc, dec = init_procedural(Encoder(cfg)).to("cuda:0"), init_procedural(Decoder(cfg)).to("cuda:0")
hot = HotPath(enc, dec)
sde = {k: v.detach().cpu() for k, v in enc.flat().items()}
sdd = {k: v.detach().cpu() for k, v in dec.flat().items()}
worst = [0.0, 0.0, 0.0]
for start in [int(a) for a in sys.argv[1:]] or [100, 300, 448, 509]:
    F = 3
    pts, pad = synthetic.frames(F, 65536, start=start)
    desc, edges, table = hot.step(pts.to("cuda:0"), pad.to("cuda:0"), (pts * 60).to("cuda:0"))
    oc, of, _ = O.encoder_forward(sde, cfg, pts, pad, fast_fps=True)
    want = torch.cat([of, oc * 60.0], dim=1)
    err = float((desc.cpu() - want).abs().max())
    for e in edges:
        R, T, conf, rmse = O.registration_forward(sdd, cfg, want[e.src], want[e.dst], 0.5)
        dT = float((e.T.cpu() - T).norm())
        M = R.double().T @ e.R.cpu().double()
        dR = float(np.arctan2(float(torch.linalg.norm(M - M.T)) / (2 * 2 ** 0.5), float((torch.trace(M) - 1) / 2)))
        worst = [max(worst[0], err), max(worst[1], dT), max(worst[2], dR)]
        flag = "" if (dT < 1e-4 and dR < 1e-4 and err < 5e-4) else "   <-- MISMATCH"
        print(f"start {start} edge {e.src}->{e.dst}: descriptor err {err:.2e}, dT {dT:.2e} m, dR {dR:.2e} rad, inliers {e.conf.numel()} vs {conf.numel()}{flag}")
print(f"worst: descriptor {worst[0]:.2e}, dT {worst[1]:.2e}, dR {worst[2]:.2e}; max |coordinate| of the last frames {float(pts[:, :3].abs().max()):.1f}")
