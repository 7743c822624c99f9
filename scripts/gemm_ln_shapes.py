"""Which (rows, Cin, Cout) does one step send through linear_layernorm / linear, and what does each cost alone?
Records the calls of one unpipelined step, then times every distinct shape: fused GEMM+LN kernel against the GEMM
kernel followed by the LayerNorm kernel (DPM_NO_FUSED_LN path)."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops, synthetic
from deeppointmap_amd.pipeline import HotPath

from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural

dev = torch.device("cuda")
cfg = default_args()
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
F = 64
pts, pad = synthetic.frames(F, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd_m = (pts * synthetic.COOR_SCALE).contiguous()
calls = collections.Counter()
orig_ln, orig_lin = ops.linear_layernorm, ops.linear


def rec_ln(x, W, *a, **k):
    calls[("ln", x.numel() // x.shape[-1], W.shape[1], W.shape[0])] += 1
    return orig_ln(x, W, *a, **k)


def rec_lin(x, W, *a, **k):
    calls[("lin", x.numel() // x.shape[-1], W.shape[1], W.shape[0])] += 1
    return orig_lin(x, W, *a, **k)


hot.step(pts, pad, pcd_m, materialize=False)
torch.cuda.synchronize()
ops.linear_layernorm, ops.linear = rec_ln, rec_lin
hot.step(pts, pad, pcd_m, materialize=False)
torch.cuda.synchronize()
ops.linear_layernorm, ops.linear = orig_ln, orig_lin


def tm(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = collections.Counter()
for (kind, R, Cin, Cout), n in sorted(calls.items(), key=lambda kv: -kv[0][1] * kv[0][2] * kv[0][3] * kv[1]):
    x = torch.randn(R, Cin, device=dev)
    W = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5
    b, g, be = torch.randn(Cout, device=dev), torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    fl = 2.0 * R * Cin * Cout
    if kind == "ln":
        us_f = tm(lambda: ops.linear_layernorm(x, W, b, g, be, act=ops.ACT_RELU))
        us_u = tm(lambda: ops.layernorm(ops.linear(x, W, b), g, be, act=ops.ACT_RELU))
        print(f"linear+LN  x{n:2d}  R={R:8d} {Cin:4d}->{Cout:4d}: fused {us_f:7.1f} us ({fl / us_f / 1e6:5.1f} TF/s)   "
              f"gemm then LN {us_u:7.1f} us")
        tot["fused"] += n * us_f
        tot["unfused"] += n * us_u
    else:
        us = tm(lambda: ops.linear(x, W, b))
        print(f"linear     x{n:2d}  R={R:8d} {Cin:4d}->{Cout:4d}: {us:7.1f} us ({fl / us / 1e6:5.1f} TF/s)")
        tot["linear"] += n * us
print({k: round(v / 1e3, 3) for k, v in tot.items()}, "ms per step")
