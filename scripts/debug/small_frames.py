"""debug: the shipped configuration on frames with FEWER points than its levels sample (4096 / 1024 / 256 / 64 / 16): the
reference pads with -1 picks / masks; against the oracle."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
from oracle import dpm_oracle as O
torch.set_grad_enabled(False)
cfg = default_args(); enc = init_procedural(Encoder(cfg)).to("cuda:0")
sd = {k: v.detach().cpu() for k, v in enc.flat().items()}
for N, lens in [(5000, [5000, 3000]), (4096, [4096, 700]), (2000, [2000, 150]), (300, [300, 40]), (64, [64, 17]), (20, [20, 3])]:
    try:
        pts, pad = synthetic.frames(2, N)
        for b, L in enumerate(lens):
            pad[b, L:] = True; pts[b, :, L:] = 0
        coor, fea, mask = enc(pts, pad)
        oc, of, om = O.encoder_forward(sd, cfg, pts, pad)
        valid = ~om
        e = float(((fea.cpu() - of).abs() * valid.unsqueeze(1)).max())
        print(N, lens, "key points equal", bool(torch.equal(coor.cpu(), oc)), "mask equal", bool(torch.equal(mask.cpu(), om)), f"feature err on valid key points {e:.2e}, on all {float((fea.cpu() - of).abs().max()):.2e}")
    except Exception as ex:
        print(N, lens, type(ex).__name__, str(ex)[:300])
