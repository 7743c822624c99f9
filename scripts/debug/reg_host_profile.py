"""debug: where does the host time of one eager registration_forward(256 x 256) go?"""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.weights import init_procedural
dec = init_procedural(Decoder(default_args())).to('cuda')
dec.graph_min_hits = 0
g = torch.Generator().manual_seed(0)
s = torch.cat([torch.rand(128, 256, generator=g), 60 * torch.randn(3, 256, generator=g)]).cuda()
d = torch.cat([torch.rand(128, 256, generator=g), 60 * torch.randn(3, 256, generator=g)]).cuda()
for _ in range(5): dec.registration_forward(s, d, num_sample=0.5)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): dec.registration_forward(s, d, num_sample=0.5)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
