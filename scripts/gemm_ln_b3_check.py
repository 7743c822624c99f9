"""Fused GEMM + LayerNorm: bf16x3 kernel (gemm_ln_b3_kernel) against the fp32-MFMA kernel (gemm_ln_kernel) -- time and error vs fp64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import knobs, ops
torch.manual_seed(0)
dev = "cuda"


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("| R | K | N | fp32 MFMA fused us | bf16x3 fused us | bf16x3 GEMM + LayerNorm us | max err fp32 | max err bf16x3 | fused == split |\n|---|---|---|---|---|---|---|---|---|")
for R, K, N in [(32768, 256, 256), (16384, 256, 256), (65536, 64, 256), (262144, 32, 128), (65536, 64, 128), (16384, 128, 128),
                (65536, 256, 64), (262144, 128, 32), (16384, 512, 128)]:
    x = torch.randn(R, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
    b, gm, bt = (torch.randn(N, device=dev) for _ in range(3))
    pre, post = torch.randn(R, N, device=dev), torch.randn(R, N, device=dev)
    n = 1024
    z = torch.nn.functional.layer_norm(x[:n].double() @ W.double().t() + b.double() + pre[:n].double(), (N,), gm.double(), bt.double(), 1e-5) + post[:n].double()
    res = {}
    for b3 in (False, True):
        knobs.GEMM_BF16X3 = knobs.GEMM_LN_BF16X3 = b3
        y = ops.linear_layernorm(x, W, b, gm, bt, pre=pre, post=post)
        res[b3] = (timed(lambda: ops.linear_layernorm(x, W, b, gm, bt, pre=pre, post=post)), float((y[:n].double() - z).abs().max()), y)
    knobs.FUSED_LN = False
    ys = ops.linear_layernorm(x, W, b, gm, bt, pre=pre, post=post)
    ts = timed(lambda: ops.linear_layernorm(x, W, b, gm, bt, pre=pre, post=post))
    knobs.FUSED_LN = True
    print(f"| {R} | {K} | {N} | {res[False][0]:.1f} | {res[True][0]:.1f} | {ts:.1f} | {res[False][1]:.2e} | {res[True][1]:.2e} | {torch.equal(ys, res[True][2])} |")
