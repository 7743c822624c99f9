"""Replay of one fuzz_encoder case against the oracle: enc_case.py B N start L0 L1 ...  (reduced configuration)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import reduced_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
from oracle import dpm_oracle as O
torch.set_grad_enabled(False)
B, N, start = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lens = [int(a) for a in sys.argv[4:4 + B]]
cfg = reduced_args()
enc = init_procedural(Encoder(cfg)).to("cuda:0")
sd = {k: v.detach().cpu() for k, v in enc.flat().items()}
pts, pad = synthetic.frames(B, N, start=start)
for b, L in enumerate(lens):
    if L < N:
        pad[b, L:] = True
        pts[b, :, L:] = 0.0
coor, fea, mask = enc(pts, pad)
oc, of, _ = O.encoder_forward(sd, cfg, pts, pad, fast_fps=False)
print("key points equal", torch.equal(coor.cpu(), oc))
d = (fea.cpu() - of).abs()
for b in range(B):
    e = d[b]
    tok = int(e.max(0).values.argmax())
    print(f"frame {b}: max err {float(e.max()):.3e} at token {tok}; tokens with err > 1e-4: {int((e.max(0).values > 1e-4).sum())} of {e.shape[1]}")
ref = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_case_oracle.pt")
if os.path.exists(ref):   # the oracle's output for the same case computed in the build container (where it is pinned to the reference)
    r = torch.load(ref)
    print("oracle here vs oracle of the build container:", float((of - r["of"]).abs().max()), "key points", torch.equal(oc, r["oc"]))
    print("HIP vs oracle of the build container:", float((fea.cpu() - r["of"]).abs().max()))
