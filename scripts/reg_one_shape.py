import sys, time, torch
sys.path.insert(0, '.')
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.weights import init_procedural
dec = init_procedural(Decoder(default_args())).to('cuda')
g = torch.Generator().manual_seed(0)
M, N = int(sys.argv[1]), int(sys.argv[2])
s = torch.cat([torch.rand(128, M, generator=g), 60 * torch.randn(3, M, generator=g)]).cuda()
d = torch.cat([torch.rand(128, N, generator=g), 60 * torch.randn(3, N, generator=g)]).cuda()
for _ in range(2): dec.registration_forward(s, d, num_sample=0.5)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): R, T, c, rmse = dec.registration_forward(s, d, num_sample=0.5)
torch.cuda.synchronize()
print(f'registration_forward {M}x{N}: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms')
