"""Wall time of ONE registration_forward call (host call to python floats, as the reference's odometry / mapping / loop
threads make it) at the three shapes of a SLAM step: eager launches against the captured HIP graph, results compared."""
import sys, time, torch
sys.path.insert(0, '.')
from deeppointmap_amd import knobs
knobs.apply_env()
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.weights import init_procedural
dec = init_procedural(Decoder(default_args())).to('cuda')
g = torch.Generator().manual_seed(0)


def timed(s, d, n=10):
    for _ in range(10):   # (the HIP runtime grows a pool once after a few hundred launches of a process: a 40 ms call)
        out = dec.registration_forward(s, d, num_sample=0.5)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        out = dec.registration_forward(s, d, num_sample=0.5)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, out


for M, N in [(256, 256), (4096, 256), (4096, 4096)]:
    s = torch.cat([torch.rand(128, M, generator=g), 60 * torch.randn(3, M, generator=g)]).cuda()
    d = torch.cat([torch.rand(128, N, generator=g), 60 * torch.randn(3, N, generator=g)]).cuda()
    dec.graph_min_hits = 0
    te, oe = timed(s, d)
    dec.graph_min_hits = 2
    tg, og = timed(s, d)
    same = all(torch.equal(a, b) for a, b in zip(oe[:3], og[:3])) and oe[3] == og[3]
    captured = any(v["graph"] is not None for k, v in dec._graphs.items() if k[:2] == (M, N))
    print(f'registration_forward {M}x{N}: eager {te:.2f} ms | graph {tg:.2f} ms (captured: {captured}, identical: {same})  '
          f'k={oe[2].shape[0]} inliers, rmse {oe[3]:.3f}')
