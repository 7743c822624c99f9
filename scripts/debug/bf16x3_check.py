"""debug: bf16x3 GEMM vs the fp32-MFMA GEMM vs fp64, and timing on the decoder's shapes"""
import ctypes, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeppointmap_amd import ops, _lib
lib = _lib.load()
dev = 'cuda:0'
torch.manual_seed(0)
def split(W):
    planes = torch.empty(3, *W.shape, device=dev, dtype=torch.int16)
    _lib.check(lib.dpm_split_bf16x3(ops._ptr(W), W.numel(), ops._ptr(planes), ops._stream(W)), "split")
    return planes
def lin3(x, Wp, b, res=None, act=0):
    R, K = x.shape; N = Wp.shape[1]
    out = torch.empty(R, N, device=dev)
    _lib.check(lib.dpm_linear_bf16x3(ops._ptr(x), K, ops._ptr(Wp), K, N * K, ops._ptr(b), ops._ptr(res), N if res is not None else 0, ops._ptr(out), N, R, K, N, act, ops._stream(x)), "lin3")
    return out
for R, K, N in [(1000, 64, 128), (32768, 256, 768), (32768, 256, 256), (4097, 512, 132), (300, 1024, 64)]:
    x = torch.randn(R, K, device=dev) * 3; W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev); res = torch.randn(R, N, device=dev)
    Wp = split(W)
    # the split is exact
    pl = (Wp.to(torch.int32) << 16).view(torch.float32) if False else None
    hi = (Wp[0].to(torch.int32) << 16).view(torch.float32); mid = (Wp[1].to(torch.int32) << 16).view(torch.float32); lo = (Wp[2].to(torch.int32) << 16).view(torch.float32)
    assert torch.equal((hi.double() + mid.double() + lo.double()).float(), W), "split not exact"
    y3 = lin3(x, Wp, b, res, 1)
    y32 = ops.linear(x, W, b, act=1, residual=res)
    ref = (x.double() @ W.double().T + b.double() + res.double()).relu()
    scale = float(ref.abs().max())
    print(f"{R}x{K}->{N}: bf16x3 vs fp64 {float((y3.double()-ref).abs().max())/scale:.2e}  fp32-mfma vs fp64 {float((y32.double()-ref).abs().max())/scale:.2e}  bf16x3 vs fp32 {float((y3-y32).abs().max())/scale:.2e}")
def tm(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for R, K, N in [(32768, 256, 768), (32768, 256, 256), (16384, 256, 768), (262144, 32, 128), (262144, 128, 32)]:
    x = torch.randn(R, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    Wp = split(W)
    t3, t32 = tm(lambda: lin3(x, Wp, b)), tm(lambda: ops.linear(x, W, b))
    fl = 2 * R * K * N
    print(f"{R}x{K}->{N}: bf16x3 {t3:.1f} us ({fl/t3/1e6:.0f} TFLOP/s-equivalent), fp32 mfma {t32:.1f} us ({fl/t32/1e6:.0f} TFLOP/s)")
