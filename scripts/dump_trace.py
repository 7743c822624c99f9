"""debug helper: run the HIP encoder on synthetic frame 0 and dump its per-stage trace."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
enc = init_procedural(Encoder(default_args())).to('cuda:0')
p = synthetic.frame(0).unsqueeze(0)
tr = {}
coor, fea, mask = enc(p, torch.zeros(1, p.shape[2], dtype=torch.bool), trace=tr)
out = {k: v.cpu().numpy() for k, v in tr.items()}
out['fea'] = fea.cpu().numpy()
np.savez_compressed('gpurun_out/trace_syn0.npz', **out)
print({k: v.shape for k, v in out.items()})
