import sys, time, torch
sys.path.insert(0, '.')
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.weights import init_procedural
dec = init_procedural(Decoder(default_args())).to('cuda')
dec.graph_min_hits = 0
g = torch.Generator().manual_seed(0)
s = torch.cat([torch.rand(128, 256, generator=g), 60 * torch.randn(3, 256, generator=g)]).cuda()
d = torch.cat([torch.rand(128, 256, generator=g), 60 * torch.randn(3, 256, generator=g)]).cuda()
ts = []
for i in range(40):
    torch.cuda.synchronize(); t = time.perf_counter()
    dec.registration_forward(s, d, num_sample=0.5)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print(" ".join(f"{x:.2f}" for x in ts))
