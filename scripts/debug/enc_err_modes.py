"""debug: encoder feature error against the oracle (reduced configuration, random frames) in the three grouping-layer forms:
plain (relative coordinates per row), folded, folded + centred.  python scripts/debug/enc_err_modes.py [passes]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeppointmap_amd import knobs, synthetic
from deeppointmap_amd.config import reduced_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
from oracle import dpm_oracle as O
torch.set_grad_enabled(False)
cfg = reduced_args()
enc = init_procedural(Encoder(cfg)).to("cuda:0")
sd = {k: v.detach().cpu() for k, v in enc.flat().items()}
rng = random.Random(7)
errs = {"plain": [], "folded": [], "centred": [], "centred r>=0.1": [], "centred r>=0.2": [], "centred r>=0.4": []}
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    B, N, start = rng.randint(1, 3), rng.randint(600, 6000), rng.randint(0, 10_000)
    pts, pad = synthetic.frames(B, N, start=start)
    _, of, _ = O.encoder_forward(sd, cfg, pts, pad, fast_fps=False)
    for mode, (f, c, rmin) in (("plain", (False, False, 0.0)), ("folded", (True, False, 0.0)), ("centred", (True, True, 0.0)),
                               ("centred r>=0.1", (True, True, 0.09)), ("centred r>=0.2", (True, True, 0.19)), ("centred r>=0.4", (True, True, 0.39))):
        knobs.FOLD_GATHER, knobs.CENTRED_GATHER, knobs.FOLD_MIN_RADIUS = f, c, rmin
        _, fea, _ = enc(pts, pad)
        errs[mode].append(float((fea.cpu() - of).abs().max()))
for mode, e in errs.items():
    e = sorted(x for x in e if x < 3e-4)   # (passes where the host's oracle flips a neighbour set excluded)
    print(f"{mode}: {len(e)} passes, median {e[len(e) // 2]:.2e}, 90 % {e[int(len(e) * 0.9)]:.2e}, max {e[-1]:.2e}")
