"""Which GEMM shapes one step issues (rows, K, N, count) and what each costs alone: the table that says where the
MFMA time of the path goes."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import ops, synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural

dev = torch.device("cuda:0")
cfg = default_args()
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
pts, pad = synthetic.frames(64, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd = (pts * 60).contiguous()
hot.step(pts, pad, pcd, materialize=False)
seen = collections.Counter()
real = ops.linear
def spy(x, W, *a, **k):
    seen[(x.numel() // x.shape[-1], x.shape[-1], W.shape[0], bool(k.get("relu", False)), k.get("residual") is not None)] += 1
    return real(x, W, *a, **k)
ops.linear = spy
import deeppointmap_amd.encoder as E, deeppointmap_amd.decoder as D
hot.step(pts, pad, pcd, materialize=False)
ops.linear = real
torch.cuda.synchronize()
tot = 0.0
rows = []
for (R, K, N, relu, res), cnt in seen.items():
    x = torch.randn(R, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(R, N, device=dev)
    for _ in range(2): real(x, W, b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): real(x, W, b, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    rows.append((us * cnt, R, K, N, cnt, us, 2 * R * K * N / us / 1e6))
    tot += us * cnt
for t, R, K, N, cnt, us, tf in sorted(rows, reverse=True):
    print(f"R={R:7d} K={K:5d} N={N:5d} x{cnt:2d}  {us:8.1f} us each  {tf:6.1f} TFLOP/s  {t:8.1f} us total ({100 * t / tot:4.1f} %)")
print(f"sum {tot / 1e3:.2f} ms")
