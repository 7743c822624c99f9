"""One 64-frame encoder pass a few times (for profilers / PMC passes of the encoder's kernels): python scripts/encode_one.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
enc = init_procedural(Encoder(default_args())).to(dev)
pts, pad = synthetic.frames(64, 65536)
pts, pad = pts.to(dev), pad.to(dev)
for _ in range(reps):
    enc(pts, pad, descriptor_scale=60.0)
torch.cuda.synchronize()
