import sys, torch; sys.path.insert(0,'/root/repo')
from deeppointmap_amd import synthetic, ops
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
from oracle import dpm_oracle as O
torch.set_grad_enabled(False)
cfg = default_args(); enc = init_procedural(Encoder(cfg)).to("cuda:0")
sd = {k: v.detach().cpu() for k, v in enc.flat().items()}
for N in (70000, 100000, 131072, 200000):
    try:
        pts, pad = synthetic.frames(1, N)
        coor, fea, mask = enc(pts, pad)
        oc, of, _ = O.encoder_forward(sd, cfg, pts, pad, fast_fps=True)
        print(N, "ok: key points equal", bool(torch.equal(coor.cpu(), oc)), "feature err", float((fea.cpu()-of).abs().max()))
    except Exception as e:
        print(N, type(e).__name__, str(e)[:200])
