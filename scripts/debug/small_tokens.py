"""debug: the decoder on very few tokens and on large loop-detection batches, against the oracle."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.weights import init_procedural
from oracle import dpm_oracle as O
torch.set_grad_enabled(False)
cfg = default_args(); dec = init_procedural(Decoder(cfg)).to("cuda:0")
sd = {k: v.detach().cpu() for k, v in dec.flat().items()}
g = torch.Generator().manual_seed(2)
def desc(*shape):
    d = torch.rand(*shape, generator=g); d[..., 128:, :] = (d[..., 128:, :] - 0.5) * 80; return d
for M, N in [(8, 5), (1, 1), (2, 64), (33, 33), (64, 1), (300, 2)]:
    s, d = desc(131, M), desc(131, N)
    try:
        R, T, conf, rmse = dec.registration_forward(s, d, num_sample=0.5)
        got = f"inliers {conf.numel()} rmse {rmse:.4f} T {T.flatten().tolist()}"
    except Exception as e:
        got = f"{type(e).__name__}: {str(e)[:80]}"
    try:
        Ro, To, co, ro = O.registration_forward(sd, cfg, s, d, 0.5)
        want = f"inliers {co.numel()} rmse {ro:.4f} T {To.flatten().tolist()}"
    except Exception as e:
        want = f"{type(e).__name__}: {str(e)[:80]}"
    print(f"M {M} N {N}:\n   hip    {got}\n   oracle {want}")
for C, M, N in [(64, 256, 256), (1, 256, 256), (200, 64, 32), (5, 1, 7)]:
    S, D = desc(C, 131, M), desc(C, 131, N)
    p, po = dec.loop_detection_forward(S, D).cpu(), O.loop_detection_forward(sd, cfg, S, D)
    print(f"loop C {C} M {M} N {N}: max diff {float((p - po).abs().max()):.2e}")
